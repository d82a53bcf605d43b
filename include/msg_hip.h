/*
 * msg_hip.h -- C ABI of libmsg_hip.so: the MI355X (gfx950) kernels behind the
 * Multi-StyleGAN generator/discriminator hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it is named host_*;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it,
 *     nothing here synchronises, allocates or keeps state between calls;
 *   - inputs are borrowed, outputs are caller-allocated and fully overwritten
 *     unless the entry says "accumulates";
 *   - return value: MSG_OK (0) or a negative MSG_E* code -- an unsupported
 *     configuration is an error, never a silent no-op (the reference silently
 *     launches nothing for unmatched modes: op_static/upfirdn2d_kernel.cu:172-211);
 *   - dtype: MSG_F32 or MSG_BF16 storage; arithmetic is always fp32.  The two entries that replace the
 *     reference's CUDA modules (msg_upfirdn2d[_pitched], msg_fused_bias_act) take every type of
 *     AT_DISPATCH_FLOATING_TYPES_AND_HALF (op_static/upfirdn2d_kernel.cu:225, op_static/fused_bias_act_kernel.cu:79):
 *     also MSG_F16 (`half`; fp32 arithmetic) and MSG_F64 (`double`: storage AND arithmetic in float64, and the `fir` /
 *     `bias` operands then point to float64 values too, as the reference's kernel / bias tensors share the input's
 *     dtype -- the precision torch.autograd.gradcheck needs).  msg_bias_act_backward takes MSG_F16 as well.
 *
 * Each entry cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef MSG_HIP_H
#define MSG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { MSG_OK = 0, MSG_EINVAL = -1, MSG_EUNSUPPORTED = -2, MSG_ELAUNCH = -3 };
enum { MSG_F32 = 0, MSG_BF16 = 1, MSG_F16 = 2, MSG_F64 = 3, MSG_F32_SPLIT = 4 };
/* MSG_F32_SPLIT (msg_conv2d_fprop* / msg_conv2d_wgrad* only): fp32 STORAGE like MSG_F32, but every product of the contraction
 * is taken as SIX bf16 MFMA products with fp32 accumulation: x = xh + xm + xl, w = wh + wm + wl (xh = bf16(x), xm = bf16(x - xh),
 * xl = bf16(x - xh - xm): all 24 mantissa bits), x w ~ xh wh + xh wm + xm wh + xh wl + xl wh + xm wm -- everything down to
 * 2^-16 of the leading product; the error is at fp32-rounding level (the exact-fp32 MFMA of MSG_F32 runs at a sixteenth of the
 * bf16 matrix rate, this at a sixth).  Workspaces and layouts are those of MSG_F32.  (A three-product form on (hi, lo) splits
 * was built too: 16 mantissa bits per operand move the training step's gradients by up to 1e-2 -- not a parity path; removed.) */

/* Library/ABI version and the code-object architecture it was built for ("gfx950").  A caller compiled against this header
 * compares msg_abi_version() with MSG_ABI_VERSION before its first launch (workspace sizes and argument lists change with it):
 *   3  deterministic reductions (workspace arguments of msg_conv2d_wgrad, msg_bias_act_backward)
 *   4  msg_scale_reduce_channels / msg_scale_bias_act removed; msg_rgb_skip_merge(+_backward) and MSG_F32_SPLIT added;
 *      msg_affine_warp's backward workspace is B*C*H*W + 1 words (the last one is the poison word)
 *   5  round-5 entries (marked "ABI 5" below) */
#define MSG_ABI_VERSION 5
int msg_abi_version(void);
const char* msg_build_arch(void);
/* Text for a MSG_E* code. */
const char* msg_strerror(int code);

/* ---------------------------------------------------------------------------
 * a1  upfirdn2d -- replaces upfirdn2d_cuda.upfirdn2d
 *     (multi_stylegan/op_static/upfirdn2d.cpp:12-19 -> upfirdn2d_op,
 *      multi_stylegan/op_static/upfirdn2d_kernel.cu:140-272).
 * x   [major, in_h, in_w, minor]   (NCHW planes: major=B*C, minor=1;
 *                                   channels-last: major=B, minor=C)
 * fir [kh, kw] float32 (un-flipped, exactly what the reference passes)
 * y   [major, out_h, out_w, minor], out = (in*up + pad0 + pad1 - k) / down + 1
 * Zero-insert by up, pad (negative = crop), true convolution with fir, keep
 * every down-th sample.  Any up/down/pad/k combination is accepted (fast
 * paths: k<=4x4 with (up,down) in {(1,1),(2,1),(1,2)} and minor % vec == 0).
 * ------------------------------------------------------------------------- */
int msg_upfirdn2d(const void* x, const float* fir, void* y, int dtype,
                  int major, int in_h, int in_w, int minor, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y,
                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* msg_upfirdn2d whose input pixels are `in_pitch` (>= minor) elements apart: a channel-slice of a wider channels-last
 * map, e.g. the gradient of one piece of a channel concatenation, filtered in place instead of being compacted first. */
int msg_upfirdn2d_pitched(const void* x, const float* fir, void* y, int dtype,
                          int major, int in_h, int in_w, int minor, int in_pitch, int kh, int kw,
                          int up_x, int up_y, int down_x, int down_y,
                          int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* msg_upfirdn2d_pitched whose OUTPUT pixels are `out_pitch` (>= minor) elements apart as well: the result is written
 * straight into its channel-slice of the map it is concatenated into (u_net_2d_discriminator.py:128-131 in the
 * reference: torch.cat of the upsampled features and the encoder's skip features). */
int msg_upfirdn2d_pitched2(const void* x, const float* fir, void* y, int dtype,
                           int major, int in_h, int in_w, int minor, int in_pitch, int out_pitch, int kh, int kw,
                           int up_x, int up_y, int down_x, int down_y,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* msg_upfirdn2d for up = down = 1 and a SEPARABLE 4x4 FIR given by its factors (fir2d = fir_y fir_x^T, which is how
 * every FIR of the models is built: multi_stylegan_generator.py:244-258, u_net_2d_discriminator.py:186-203), on a
 * channels-last map (major = B, minor = C, minor % vec == 0).  Sliding-window evaluation, 8 instead of 16 taps per
 * output; results equal msg_upfirdn2d's up to fp32 re-association.  Other shapes: MSG_EUNSUPPORTED (use msg_upfirdn2d). */
int msg_upfirdn2d_separable(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                            int major, int in_h, int in_w, int minor, int kh, int kw,
                            int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* msg_upfirdn2d_separable with the activation stage of the layer that owns the blur fused behind it:
 *   y = leaky_relu(blur(x) + noise_weight[0] * noise[b or 0, pixel] + act_bias[c], alpha) * scale
 * (the upsampling StyledConv2d: transposed conv -> Blur -> NoiseInjection -> FusedLeakyReLU,
 * multi_stylegan_generator.py:267-292,329-344).  The activation sees the fp32 blur result, i.e. one rounding less than
 * msg_upfirdn2d_separable followed by msg_fused_bias_act (within one unit in the last place of the storage type). */
int msg_upfirdn2d_separable_act(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                                int major, int in_h, int in_w, int minor, int kh, int kw,
                                int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                const float* act_bias, const float* noise, const float* noise_weight,
                                int noise_batch, float alpha, float scale, void* stream);
/* ... which also leaves the SIGN BYTES of its output for the activation's backward (bf16, minor % 8 == 0; mask may be NULL):
 * mask [major * out_h * out_w][minor / 8] (tile 1 x minor in msg_bias_act_backward_mask's terms), bit e of byte
 * (pixel, c / 8) = (y[pixel][c + e] > 0) -- everything
 * FusedLeakyReLUFunctionBackward (op_static/fused_act.py:22-51) reads of `out`, at a sixteenth of its size; see
 * msg_bias_act_backward_mask. */
int msg_upfirdn2d_separable_act_mask(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                                     int major, int in_h, int in_w, int minor, int kh, int kw,
                                     int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                     const float* act_bias, const float* noise, const float* noise_weight,
                                     int noise_batch, float alpha, float scale, unsigned char* mask, void* stream);

/* ---------------------------------------------------------------------------
 * a2  fused bias + (noise) + leaky-ReLU -- replaces fused_act_cuda.fused_bias_act
 *     (multi_stylegan/op_static/fused_bias_act.cpp:11-17 -> fused_bias_act_op,
 *      multi_stylegan/op_static/fused_bias_act_kernel.cu:18-99) and fuses the
 *     NoiseInjection add in front of it (multi_stylegan_generator.py:288-292).
 * x, y   flat [size_x]; bias index of element i is (i / step_b) % size_b
 *        (NCHW: step_b = H*W; channels-last or [B,C]: step_b = 1).
 * bias   [size_b] float32 or NULL.
 * ref    flat [size_x] or NULL; with grad=1 the slope is chosen by sign(ref)
 *        (the saved forward OUTPUT) instead of sign(x+b): kernel.cu:44.
 * noise  NULL, or float32 [noise_batch(1 or B), pix] added as
 *        (*noise_weight) * noise[pixel(i)] before the bias; pixel(i) is
 *        (i / step_b / size_b) * pix + i % step_b   for NCHW  (pix = step_b)
 *        i / size_b                                  for channels-last (step_b == 1)
 *        and wraps modulo `pix` when noise_batch == 1.
 * act    3 = leaky ReLU (the only activation the reference ever uses), 1 = linear.
 * grad   0 forward, 1 first derivative (needs ref), 2 -> zeros.
 * y = act(x + w*noise + b) * scale, computed in fp32.
 * ------------------------------------------------------------------------- */
int msg_fused_bias_act(const void* x, const float* bias, const void* ref, void* y, int dtype,
                       long long size_x, int step_b, int size_b,
                       const float* noise, const float* noise_weight, int noise_batch, int pix,
                       int act, int grad, float alpha, float scale, void* stream);

/* ---------------------------------------------------------------------------
 * a2  backward of the above with its reductions -- replaces
 *     FusedLeakyReLUFunctionBackward.forward (multi_stylegan/op_static/fused_act.py:24-42:
 *     one fused_bias_act(grad=1) call + grad_input.sum(dim) in PyTorch).
 * gx = gy * scale * (out > 0 ? 1 : alpha);  grad_bias[c] = sum gx;
 * grad_noise_weight[0] = sum gx * noise[pixel]   (if noise != NULL).
 * grad_bias / grad_noise_weight are float32 and OVERWRITTEN.  The sums are deterministic: workgroups leave partial sums
 * in `ws` (float32, at least msg_bias_act_backward_workspace(...) elements, contents irrelevant) and a second launch adds
 * them in index order -- no float atomics, bit-identical results for identical inputs.  ws may be NULL when neither sum
 * is requested.
 * ------------------------------------------------------------------------- */
long long msg_bias_act_backward_workspace(long long size_x, int step_b, int size_b, int has_noise);
int msg_bias_act_backward(const void* gy, const void* out, void* gx, int dtype,
                          long long size_x, int step_b, int size_b,
                          float* grad_bias, const float* noise, float* grad_noise_weight,
                          int noise_batch, int pix, float alpha, float scale,
                          float* ws, long long ws_floats, void* stream);
/* sums[c] = sum over the pixels of a channels-last map x [size_x / C][C]: the bias gradient of a convolution with no activation
 * behind it (`F.conv2d(..., bias)` of the discriminator's strided convs, multi_stylegan/equalized_layer.py:63-74).  fp32,
 * overwritten, deterministic (workspace and second stage of msg_bias_act_backward: ws_floats >=
 * msg_bias_act_backward_workspace(size_x, 1, C, 0)).  MSG_F32 / MSG_BF16, C a whole number of 16-byte vectors. */
int msg_channel_sums(const void* x, float* sums, int dtype, long long size_x, int C, float* ws, long long ws_floats,
                     void* stream);
/* msg_bias_act_backward for a channels-last bf16 map (step_b = 1, size_b % 8 == 0) whose forward launch left the sign
 * bytes of its output (msg_conv2d_fprop_act_mask / msg_upfirdn2d_separable_act_mask): `mask` replaces `out` -- the pass
 * moves 2 1/16 instead of 3 maps.  The bytes lie in the PRODUCER's tile order: tiles of tile_m consecutive pixels x tile_n
 * channels one after the other, [tile_m][tile_n / 8] bytes each, pixel tiles outermost (a workgroup of the producer writes one
 * contiguous block); tile 1 x size_b is the plain [pixel][size_b / 8] map.  Same results bit for bit (the bytes are the test
 * `out > 0` on the stored values), same workspace (msg_bias_act_backward_workspace(size_x, 1, size_b, has_noise)). */
int msg_bias_act_backward_mask(const void* gy, const unsigned char* mask, int tile_m, int tile_n, void* gx, int dtype,
                               long long size_x, int size_b,
                               float* grad_bias, const float* noise, float* grad_noise_weight,
                               int noise_batch, int pix, float alpha, float scale,
                               float* ws, long long ws_floats, void* stream);
/* (ABI 5) out[c] = sum_r part[r][cols], r in index order (fp32, deterministic): the second stage of the partial-sum reductions as an
 * entry of its own -- e.g. the [groups][B][I] style-gradient partials of msg_modulate_backward. */
int msg_sum_rows(const float* part, float* out, long long rows, int cols, void* stream);
/* (ABI 5) msg_bias_act_backward_mask for the output of a styled layer that ALSO feeds the level's image head -- a 1x1 modulated
 * conv without demodulation to n_head <= 8 planes (reference: OutputBlock, multi_stylegan_generator.py:513-523, reading the
 * StyledConv2d output of :384-411).  The head's data gradient is formed inside this pass instead of being written as a full map
 * and summed by autograd:
 *     gx = (gy + h) * scale * (out > 0 ? 1 : alpha),   h[q][c] = wscale * style[b][c] * sum_o ghead[q][o] * whead[o][c]
 * gy: the other consumer's gradient (bf16 [pixels][size_b]) or NULL; ghead: bf16 [pixels][8], planes >= n_head padding; whead fp32
 * [n_head][size_b]; style fp32 [pixels / pix][size_b]; pix = pixels per sample (a workgroup's pixel range must lie inside one
 * sample: MSG_EUNSUPPORTED otherwise, as for anything but bf16).  Sums, workspace and determinism as msg_bias_act_backward_mask. */
int msg_bias_act_backward_mask_head(const void* gy, const void* ghead, const float* whead, const float* style,
                                    float wscale, int n_head, const unsigned char* mask, int tile_m, int tile_n,
                                    void* gx, int dtype, long long size_x, int size_b,
                                    float* grad_bias, const float* noise, float* grad_noise_weight,
                                    int noise_batch, int pix, float alpha, float scale,
                                    float* ws, long long ws_floats, void* stream);

/* ---------------------------------------------------------------------------
 * a3/a4  dense contractions on the matrix cores (channels-last, implicit GEMM).
 * The reference has no native boundary here: it calls F.conv2d / F.conv_transpose2d with groups = batch
 * (multi_stylegan/multi_stylegan_generator.py:391-411) and F.conv2d / F.linear for the equalized layers
 * (multi_stylegan/equalized_layer.py:63-74, 244-254).  These two entries sit behind ModulatedConv2d.forward,
 * EqualizedConv2d.forward and EqualizedLinear.forward.
 *
 * msg_conv2d_fprop:
 *   y[b,oh,ow,n] = bias[n] + sum_{kh,kw,c} x[b, (oh*stride+kh-pad)/in_up, (ow*stride+kw-pad)/in_up, c] * w[(b)][n][kh][kw][c]
 *   x  [B, IH, IW, Cx]   Cx = channel stride (multiple of 16 B); channels >= the real count must hold finite values
 *   w  [(B)][N][kh*kw][Ck]  Ck = input channels zero-padded to a multiple of 128 B; w_batch_stride = 0 -> shared
 *   y  [B, OH, OW, ldy]  (pixel_shuffle = 1: N = 4*O and y is [B, 2*OH, 2*OW, ldy] with
 *                         y[b, 2oh+dy, 2ow+dx, o] = result[n = (2dy+dx)*O + o] -- the 2x2 stride-2 transposed conv)
 *   in_up > 1 (stride must be 1): samples exist only where the numerator is divisible (transposed strided conv).
 *   Data gradients are the same entry with the caller's re-laid weights.
 * msg_conv2d_wgrad:
 *   gw[(b)][o][tap][i] (+)= sum_{pixels} gy[b,oh,ow,o] * x[b, oh*stride+kh-pad, ow*stride+kw-pad, i]
 *   gw fp32 [(B)][O][kh*kw][ldgw] (ldgw % 4 == 0), OVERWRITTEN; per_sample = 1: one result per sample (its pixels split
 *   into k_chunks K-slices), otherwise ONE result for the whole batch (the library folds the batch into K and chooses the
 *   number of K-slices itself; k_chunks is ignored unless folding is impossible).
 *   A result that is the sum of several K-slices is formed deterministically: every slice stores its partial result into
 *   its own slab of `ws` (float32, at least msg_conv2d_wgrad_workspace(...) elements -- 0 when nothing is split; contents
 *   irrelevant) and a second launch adds the slabs in slice order.  No float atomics: identical inputs give bit-identical
 *   gradients (the reference's cuDNN weight gradients are not deterministic; its training step is what ours must match).
 *   pixel_shuffle = 1: OH,OW are the LOW-res extent, gy is [B,2*OH,2*OW,ldgy], tap = (dy,dx).
 *   oi_major = 1 writes gw[(b)][o][i][tap] (the parameter's own [O,I,kh,kw] layout) instead; gain multiplies the result.
 * ------------------------------------------------------------------------- */
int msg_conv2d_fprop(const void* x, const void* w, const float* bias, void* y, int dtype,
                     int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                     int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                     long long w_batch_stride, void* stream);
int msg_conv2d_wgrad(const void* gy, const void* x, float* gw, int dtype,
                     int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy, int O, int ldgw,
                     int kh, int kw, int stride, int pad, int pixel_shuffle,
                     int per_sample, int k_chunks, int oi_major, float gain,
                     float* ws, long long ws_floats, void* stream);
/* Workspace elements (float32) the call above needs for the same geometry; negative = MSG_E* code. */
long long msg_conv2d_wgrad_workspace(int dtype, int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy,
                                     int O, int ldgw, int kh, int kw, int stride, int pad, int pixel_shuffle,
                                     int per_sample, int k_chunks);

/* ---------------------------------------------------------------------------
 * a3  weight modulation / demodulation of the dual-styled conv (multi_stylegan/multi_stylegan_generator.py:379-388).
 * The reference does this with elementwise torch ops on a [B,O,I,kh,kw] tensor; these entries write the per-sample
 * weights directly in the contraction kernels' layouts and turn per-sample weight gradients back into parameter and
 * style gradients (including the derivative of the demodulation norm).
 *   msg_demod_coeff:       d[b,o] = rsqrt(scale^2 * sum_{i,t} (W[o,i,t]*s[b,i])^2 + eps);  W [O][I][taps], s [B][I]
 *   msg_scale_rows_cols:   out[b][r][t][c] = gain * base[r][t][c] * rowscale[b][r] * colscale[b][c], c >= C -> 0;
 *                          base fp32 [R][T][C]; rowscale/colscale fp32 or NULL (= 1); out [B][R][T][Ck] f32/bf16
 *   msg_modulate_backward: gwk fp32 [B][O][taps][ldg] (per-sample dL/dw, w = d*scale*W*s) ->
 *                          gW fp32 [O][I][taps] (overwritten) and gs_part fp32 [ceil(O/o_group)][B][I] (overwritten;
 *                          the caller sums over the first axis); d = NULL means "no demodulation".
 *                          Limits: I <= 512, taps <= 9, B <= 16 (else MSG_EUNSUPPORTED).
 * ------------------------------------------------------------------------- */
int msg_demod_coeff(const float* W, const float* s, float* d, int B, int O, int I, int taps,
                    float scale, float eps, void* stream);
int msg_scale_rows_cols(const float* base, const float* rowscale, const float* colscale, void* out,
                        int dtype, int B, int R, int T, int C, int Ck, float gain, void* stream);
/* Demodulation coefficients + forward per-sample weights in one launch (supersedes msg_demod_coeff followed by
 * msg_scale_rows_cols on the forward path): wsq [O][C] = sum over taps of W^2 (cached by the caller per weight update),
 * base [R][T][C] (R = O, or 4*O for the 2x2 transposed conv), style [B][C];
 * out[b][r][t][c] = scale * d[b][r % O] * base[r][t][c] * style[b][c],  d_out[b][o] = d (may be NULL). */
int msg_modulate_weights(const float* base, const float* wsq, const float* style, void* out, float* d_out,
                         int dtype, int B, int R, int O, int T, int C, int Ck, float scale, float eps, void* stream);
int msg_modulate_backward(const float* gwk, const float* W, const float* s, const float* d, float* gW,
                          float* gs_part, int B, int O, int I, int taps, int ldg, int o_group,
                          float scale, void* stream);
/* Second order of the same (the path-length regulariser differentiates the style gradient of the first backward once more,
 * multi_stylegan_generator.py:193-200 through :384-388): for a cotangent v [B][I] of the style gradient gs,
 *   msg_scale_rows_cols2:   out[b][r][t][c] = gain * base[r][t][c] * (row1[b][r]*col1[b][c] + row2[b][r]*col2[b][c]) -- with
 *                           (row1,col1,row2,col2) = (d, v, dd, s), dd[b][o] = -scale^2 d^3 sum_i s v wsq[o][i], the derivative of
 *                           the per-sample weight set along v (= dL2/d gwk), in the contraction kernels' layouts;
 *   msg_modulate_backward2: dL2/dW -> gW [O][I][taps] and dL2/ds -> gs_part [ceil(O/o_group)][B][I] at fixed gwk, L2 = <v, gs>
 *                           (both overwritten; d = NULL: no demodulation).  Limits: I % 4 == 0, I <= 512, taps in {1,4,9},
 *                           B <= 16, 16-byte aligned operands (else MSG_EUNSUPPORTED).                                        */
int msg_scale_rows_cols2(const float* base, const float* row1, const float* col1, const float* row2, const float* col2,
                         void* out, int dtype, int B, int R, int T, int C, int Ck, float gain, void* stream);
int msg_modulate_backward2(const float* gwk, const float* W, const float* s, const float* d, const float* v,
                           float* gW, float* gs_part, int B, int O, int I, int taps, int ldg, int o_group,
                           float scale, void* stream);

/* Every K-contiguous image the contraction / modulation kernels read from ONE parameter, in one pass (all outputs
 * optional): w [O][I][T] fp32 as the reference stores it (equalized_layer.py, multi_stylegan_generator.py:330);
 * fwd [O][T][Ck] (t_major: [T][O][Ck]) = gain*w zero-padded in i; dgrad [I][T][Ok] with the taps flipped when flip != 0,
 * zero-padded in o; wsq [O][I] = sum over taps of w^2.  fwd/dgrad in `dtype`.  T <= 16. */
int msg_relayout_weight(const float* w, void* fwd, void* dgrad, float* wsq, int dtype,
                        int O, int I, int T, int Ck, int Ok, int flip, int t_major, float gain, void* stream);

/* Tap gathering for convolutions with very few input channels (the discriminator's first layer, 6 channels):
 * y[b][h][w][t*C + c] = x[b][h + kh_t - pad][w + kw_t - pad][c] (zero outside the image / beyond taps*C), Ko channels per
 * output pixel, x with channel pitch Cx.  The kh x kw conv then runs as a 1x1 conv over the gathered map (one K-step
 * instead of kh*kw mostly-zero ones; the weight gradient reads gy once instead of once per tap). */
int msg_gather_taps(const void* x, void* y, int dtype, int B, int H, int W, int Cx, int C, int kh, int kw,
                    int pad, int Ko, void* stream);
/* ABI 5.  The adjoint of msg_gather_taps: g [B, H, W, Ko] (K index tap * C + c, as msg_gather_taps writes it) ->
 * gx [B, H, W, ldx], gx[q, c] = sum over taps of g[q - offset(tap), tap * C + c] (+ add[q, c] when add != NULL, pixel pitch
 * ld_add), fp32 accumulation, channels C .. ldx - 1 zeroed; C <= 8.  The data gradient of a few-channel 'same' convolution =
 * a 1x1 contraction to Ko channels followed by this fold (EqualizedConv2d of the first discriminator block,
 * multi_stylegan/u_net_2d_discriminator.py:161-170 with equalized_layer.py:57-74). */
int msg_fold_taps(const void* g, const void* add, void* gx, int dtype, int B, int H, int W, int Ko, int C, int kh, int kw,
                  int pad, int ldx, int ld_add, void* stream);

/* y = (a + beta*b) * gain over n elements (n multiple of the 16-byte vector, all pointers 16-B aligned): the
 * residual merges (main + residual)/sqrt(2) of multi_stylegan/u_net_2d_discriminator.py:185,381 in one pass. */
int msg_scaled_add(const void* a, const void* b, void* y, int dtype, long long n, float beta, float gain, void* stream);
/* y = (gamma[0] * a + b) * gain with gamma a DEVICE fp32 scalar -- the merge of the NonLocalBlock, (gamma * o + residual) / sqrt(2)
 * (multi_stylegan/u_net_2d_discriminator.py:381; gamma is a learnt parameter) -- and its backward in one pass:
 *   ga = gamma * gain * gy,  gb = gain * gy,  g_gamma[0] = gain * sum(gy * a)   (block partials in ws, added in a fixed order by a
 * second launch: deterministic).  Dense maps of n elements in one layout, MSG_F32 / MSG_BF16, n a multiple of the 16-byte vector. */
/* The generator's RGB skip path, one launch per level (multi_stylegan_generator.py:513-523: OutputBlock.forward):
 *   out[b,c,y,x] = float(conv[b,y,x,c]) + bias[c] + upfirdn2d(skip, fir, up = 2, pad = (2, 1))[b,c,y,x]
 * conv [B][H][W][ld] channels-last (MSG_F32 / MSG_BF16, pixel pitch ld >= C elements: the thin 1x1 conv's output), bias fp32
 * [C] or NULL, skip fp32 planes [B][C][H/2][W/2] or NULL (first level), fir DEVICE pointer to the 4 x 4 taps (required with
 * skip), out fp32 planes [B][C][H][W].  C <= 8, W % 4 == 0, H even; anything else MSG_EUNSUPPORTED.
 * msg_rgb_skip_merge_backward: g fp32 planes [B][C][H][W] -> g_conv [B][H][W][ld] (cast, padding channels zeroed; may be
 * NULL) and g_skip fp32 [B][C][H/2][W/2] (the transposed FIR as a gather: no atomics; may be NULL).  The op is linear: its
 * second-order pass is msg_rgb_skip_merge on the cotangents. */
int msg_rgb_skip_merge(const void* conv, int dtype, int ld, const float* bias, const float* skip, const float* fir,
                       float* out, int B, int C, int H, int W, void* stream);
int msg_rgb_skip_merge_backward(const float* g, void* g_conv, int dtype, int ld, float* g_skip, const float* fir,
                                int B, int C, int H, int W, void* stream);
int msg_gamma_merge(const void* a, const void* b, const float* gamma, void* y, int dtype, long long n, float gain, void* stream);
long long msg_gamma_merge_backward_workspace(void);
int msg_gamma_merge_backward(const void* gy, const void* a, const float* gamma, void* ga, void* gb, float* g_gamma, int dtype,
                             long long n, float gain, float* ws, void* stream);
/* The same for [rows][cols] operands with row pitches (elements): channel-slices of channels-last buffers, e.g. the
 * gradient of a skip connection (a slice of the concatenated map's gradient).  cols and pitches multiples of the vector. */
int msg_scaled_add_rows(const void* a, const void* b, void* y, int dtype, long long rows, int cols,
                        long long lda, long long ldb, long long ldy, float beta, float gain, void* stream);

/* Row softmax of the non-local block's attention map and its backward (u_net_2d_discriminator.py:378:
 * F.softmax(torch.bmm(theta^T, phi), -1)): x, y [rows][cols] contiguous in the storage type, fp32 arithmetic,
 *   y = softmax(x) along cols;   gx = y * (gy - sum_c gy[c] * y[c]).
 * One wave keeps a row in registers (read once, written once): cols a multiple of the 16-byte vector and at most 4096;
 * larger rows: MSG_EUNSUPPORTED. */
int msg_softmax_rows(const void* x, void* y, int dtype, long long rows, int cols, void* stream);
int msg_softmax_rows_backward(const void* y, const void* gy, void* gx, int dtype, long long rows, int cols,
                              void* stream);
/* ABI 5.  Second-order terms of the softmax backward for a cotangent v of gx (R1: the backward is differentiated again,
 * reference loss.py:311-316 through u_net_2d_discriminator.py:378), one pass over the maps:
 *   d_gy = y * (v - <v, y>),   d_y = v * (gy - <gy, y>) - gy * <v, y>     (row-wise inner products, fp32 arithmetic). */
int msg_softmax_rows_backward2(const void* y, const void* gy, const void* v, void* d_y, void* d_gy, int dtype,
                               long long rows, int cols, void* stream);

/* ---------------------------------------------------------------------------
 * a6 / 8f-1  fused attention of the NonLocalBlock -- replaces the torch.bmm -> F.softmax -> torch.bmm sequence of
 *     multi_stylegan/u_net_2d_discriminator.py:376-380 (beta = softmax(theta^T phi), o = g beta^T) without the
 *     [B, Nq, Nk] attention map ever reaching HBM.
 * q  [B, Nq, dk]  theta(x), one row per query pixel        k  [B, Nk, dk]  max-pooled phi(x), one row per key pixel
 * v  [B, Nk, dv]  max-pooled g(x)                           vt [B, dv, Nk]  its transpose (the caller provides both
 * orientations of the operands whose contraction index would otherwise be strided: qt [B,dk,Nq], kt [B,dk,Nk],
 * dOt [B,dv,Nq]); all dense, in the storage type.
 * forward:  o [B, Nq, dv] = softmax_rows(q k^T) v,  lse [B, Nq] fp32 = log sum_j exp(q_i . k_j)  (kept for backward)
 * backward: given dO [B, Nq, dv], o and lse of the forward:  dq, dk_out, dv_out  (probabilities are recomputed from q,
 *           k and lse; delta [B, Nq] fp32 is scratch the caller provides (it receives sum_d dO * o); no atomics:
 *           results are deterministic).
 * Supported: (dk, dv) = (48, 192) -- the reference's 384-channel blocks -- and (16, 64); Nq, Nk multiples of 128;
 * MSG_BF16 (bf16 MFMA, fp32 accumulate / softmax) and MSG_F32 (exact-fp32 MFMA).  Anything else:
 * MSG_EUNSUPPORTED (msg_nonlocal_attention_supported tells beforehand). */
int msg_nonlocal_attention_supported(int B, int Nq, int Nk, int dk, int dv);
int msg_nonlocal_attention_fwd(const void* q, const void* k, const void* vt, void* o, float* lse, int dtype,
                               int B, int Nq, int Nk, int dk, int dv, void* stream);
/* The 2x2 / stride-2 max-pooling in front of the attention (F.max_pool2d on phi(x) and g(x),
 * multi_stylegan/u_net_2d_discriminator.py:366-370), channels-last, MSG_BF16 / MSG_F32, H and W even, C a whole number of
 * 16-byte vectors.  x [B, H, W, C] with pixel pitch ldx >= C; y [B, H/2, W/2, C] dense; idx (NULL when no backward follows):
 * one 16-bit word per 16-byte vector of y, two bits per element = the window position that won (first maximum in scan
 * order, NaN wins: the library's rule).  Backward: gx [B, H, W, C] dense, every element written (gy at the winner, 0 elsewhere). */
int msg_maxpool2x2_fwd(const void* x, void* y, unsigned short* idx, int dtype, int B, int H, int W, int C, long long ldx,
                       void* stream);
int msg_maxpool2x2_bwd(const void* gy, const unsigned short* idx, void* gx, int dtype, int B, int H, int W, int C,
                       void* stream);
/* ABI 5.  The transpose of the backward's scatter, i.e. the derivative of msg_maxpool2x2_bwd with respect to gy for a cotangent
 * v of gx (second-order graphs, R1): v [B, H, W, C] channels-last with pixel pitch ldv, read at the positions idx names ->
 * out [B, H/2, W/2, C] dense. */
int msg_maxpool2x2_gather(const void* v, const unsigned short* idx, void* out, int dtype, int B, int H, int W, int C,
                          long long ldv, void* stream);

/* msg_nonlocal_attention_bwd_splits: in how many parts the dK / dV kernel splits its query sweep for this problem; when
 * it is more than 1 the caller passes `workspace` with splits * B * Nk * (dk + dv) floats (scratch, fully overwritten). */
int msg_nonlocal_attention_bwd_splits(int B, int Nq, int Nk);
/* (qt and dOt -- transposed copies of q and dO -- are no longer read and may be NULL: the dK / dV kernel takes Q^T and dO^T
 *  out of the row-major tiles with transposing LDS reads.  kt, the small [B, dk, Nk] copy of k, is still an operand.) */
int msg_nonlocal_attention_bwd(const void* q, const void* qt, const void* k, const void* kt, const void* v,
                               const void* dO, const void* dOt, const void* o, const float* lse, float* delta,
                               void* dq, void* dk_out, void* dv_out, float* workspace, int dtype,
                               int B, int Nq, int Nk, int dk, int dv, void* stream);

/* ---------------------------------------------------------------------------
 * 8f-2  batched affine warp of adaptive discriminator augmentation -- replaces the per-stage
 *     images[idx] = kaf.rotate(...) / kaf.apply_affine(images[idx], params, flags) of
 *     multi_stylegan/adaptive_discriminator_augmentation.py:120-199 (kornia 0.4.1) for a whole batch in one launch.
 * x, y      [B, C, H, W] fp32 (NCHW planes).
 * select_u  [B] uniform numbers; image b is transformed iff select_u[b] <= thr with thr = *p (rot_prob = 0) or
 *           1 - sqrt(1 - *p) (rot_prob = 1); p is a DEVICE scalar (the augmentation probability the controller
 *           adapts); other images are copied through.
 * angle_deg [B] degrees or NULL (then angle_const for every image); scale_xy [B, 2] or NULL (1, 1).
 * The warp is kornia's: M = [[cos, sin], [-sin, cos]] diag(scale) about (cx, cy) with the translation column
 * ((1 - m00) cx - m01 cy, m01 cx + (1 - m00) cy); output pixel -> normalised (affine_grid, align_corners) ->
 * N M^-1 N^-1 (N: pixel [0, size-1] -> [-1, 1]) -> pixel (grid_sample, align_corners), bilinear, padding 0 = zeros /
 * 2 = reflection.  (kaf.apply_affine passes -angle: the caller negates.)
 * backward != 0: x is the gradient of y, y receives the gradient of x (overwritten) -- the transpose of the bilinear
 * gather, a scatter, accumulated in 64-bit FIXED POINT (2^-38 resolution) in `workspace` (B*C*H*W + 1 8-byte words,
 * contents irrelevant) and converted by a second launch: integer addition is associative, so the result does not depend on
 * the order in which the atomics arrive (deterministic).  A contribution that is not finite or is >= 2^17 in magnitude
 * has no fixed-point image: it raises the last workspace word and the gradient of every transformed image comes out NaN
 * (a diverged gradient stays visible downstream instead of wrapping into finite values).  workspace may be NULL for
 * the forward.
 * Parity unpinned, see oracle/ada.py. */
int msg_affine_warp(const float* x, float* y, const float* angle_deg, float angle_const, const float* scale_xy,
                    const float* select_u, const float* p, int rot_prob, float cx, float cy, int padding,
                    int align_corners, int B, int C, int H, int W, int backward, void* workspace, void* stream);

/* ---------------------------------------------------------------------------
 * a5 / 8f-1  minibatch standard deviation -- replaces MinibatchStdDev.forward
 *     (multi_stylegan/u_net_2d_discriminator.py:205-217: std over the batch, mean over (c, h, w), one extra plane, cat).
 * x  channels-last [B, H, W] pixels of C channels, `ldx` elements apart;  y the same pixels with `ldy` > C channels:
 * y[..., :C] = x, y[..., C] = stat[g], further pad channels 0, where the batch is `groups` independent batches of
 * B / groups samples concatenated along dim 0 and stat[g] = mean_{c,h,w} sqrt(max(var_group(x), alpha)).
 * stat [groups] fp32 out; workspace: msg_minibatch_stddev_workspace(...) floats.  Deterministic (fixed-order sums).
 * backward: gx[..., :C] (pitch ldgx) = gy[..., :C] + gstat[g] * d stat / d x, gstat [groups] fp32 = the summed
 * gradient of each group's plane. */
long long msg_minibatch_stddev_workspace(int C, int H, int W, int groups, int dtype);
int msg_minibatch_stddev(const void* x, void* y, float* stat, float* workspace, int dtype,
                         int B, int C, int H, int W, int ldx, int ldy, int groups, float alpha, void* stream);
int msg_minibatch_stddev_backward(const void* x, const void* gy, const float* gstat, void* gx, int dtype,
                                  int B, int C, int H, int W, int ldx, int ldgy, int ldgx, int groups,
                                  float alpha, void* stream);

/* msg_conv2d_fprop with the activation stage of the layer fused into the epilogue:
 *   y = leaky_relu(conv(x, w) + noise_weight[0] * noise[b or 0, pixel] + act_bias[n], alpha) * scale
 * i.e. EqualizedConv2d -> FusedLeakyReLU (u_net_2d_discriminator.py:160-171) and ModulatedConv2d -> NoiseInjection ->
 * FusedLeakyReLU (multi_stylegan_generator.py:267-292) in one launch.  The activation is applied to the conv result
 * rounded to the storage type, so the output is bit-identical to msg_conv2d_fprop followed by msg_fused_bias_act.
 * act_bias [N] fp32 or NULL; noise [noise_batch][OH*OW] fp32 or NULL (noise_batch 1 or B); noise_weight: device scalar.
 * No in_up / pixel_shuffle forms (those layers are followed by a FIR pass before their activation). */
int msg_conv2d_fprop_act(const void* x, const void* w, void* y, int dtype,
                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                         int kh, int kw, int stride, int pad, long long w_batch_stride,
                         const float* act_bias, const float* noise, const float* noise_weight,
                         int noise_batch, float alpha, float scale, void* stream);
/* ... which also leaves the sign bytes of its output (see msg_upfirdn2d_separable_act_mask): mask, B * OH * OW * N / 8 bytes
 * in the order of the kernel's output tiles (256 pixels x 256 channels for plan 3, 128 x 128 for plan 4: the tile_m / tile_n of
 * msg_bias_act_backward_mask), or NULL.  Only the row-sharing 3x3 kernels write them (msg_conv2d_fprop_plan(...) == 3 or 4,
 * bf16): any other problem with mask != NULL is MSG_EUNSUPPORTED -- ask the plan first. */
int msg_conv2d_fprop_act_mask(const void* x, const void* w, void* y, int dtype,
                              int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                              int kh, int kw, int stride, int pad, long long w_batch_stride,
                              const float* act_bias, const float* noise, const float* noise_weight,
                              int noise_batch, float alpha, float scale, unsigned char* mask, void* stream);

/* msg_conv2d_fprop with the residual merge of a discriminator block fused into the epilogue:
 *   y = (conv(x, w) + residual) * gain        (u_net_2d_discriminator.py:185: (main + residual_mapping(x)) / sqrt(2))
 * residual: a map with the output's pixels and >= N channels, channel pitch res_ld elements, same storage type.
 * Bit-identical to msg_conv2d_fprop followed by msg_scaled_add. */
int msg_conv2d_fprop_residual(const void* x, const void* w, void* y, int dtype,
                              int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                              int kh, int kw, int stride, int pad, long long w_batch_stride,
                              const void* residual, int res_ld, float gain, void* stream);

/* (ABI 5) The DATA GRADIENT of a 3x3 'same' conv whose input was the output of a fused bias (+ noise) + leaky-ReLU stage, with
 * that stage's backward in the epilogue -- what the reference runs as conv2d_backward_input followed by
 * FusedLeakyReLUFunctionBackward (op_static/fused_act.py:24-42) on the map in between, which here is never written:
 *   y = (conv(x, w) [+ residual]) * (s > 0 ? scale : scale * alpha),  grad_bias[n] = sum_pixels y,  grad_noise_weight = sum y * noise
 * x: the gradient arriving at the conv's output, w: its data-gradient weight image (as for msg_conv2d_fprop).  s = the stage's
 * stored output, given by `sign_mask` (the bytes of msg_conv2d_fprop_act_mask / msg_upfirdn2d_separable_act_mask, tiles
 * mask_tile_m x mask_tile_n with mask_tile_m 1 or a multiple of 64, mask_tile_n a multiple of 128) or by `sign_map` (that output itself, bf16, channel pitch
 * sign_ld) -- exactly one of the two.  residual: a second gradient of the same map (added first), or NULL.  noise [noise_batch]
 * [OH*OW] fp32 with grad_noise_weight, or both NULL.  Sums: fp32, overwritten, deterministic (per-tile partials in `ws`, summed in
 * index order; ws_floats >= msg_conv2d_fprop_act_backward_workspace(...)).  Only the row-sharing kernels have this epilogue:
 * MSG_EUNSUPPORTED unless msg_conv2d_fprop_plan(...) is 3 or 4 (the workspace query then returns 0), bf16, ldy == N. */
long long msg_conv2d_fprop_act_backward_workspace(int dtype, int B, int IH, int IW, int Cx, int Ck, int OH, int OW,
                                                  int N, int kh, int kw, long long w_batch_stride, int has_noise);
int msg_conv2d_fprop_act_backward(const void* x, const void* w, void* y, int dtype,
                                  int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                  int kh, int kw, int stride, int pad, long long w_batch_stride,
                                  const void* residual, int res_ld,
                                  const unsigned char* sign_mask, int mask_tile_m, int mask_tile_n,
                                  const void* sign_map, int sign_ld, float alpha, float scale,
                                  float* grad_bias, const float* noise, int noise_batch, float* grad_noise_weight,
                                  float* ws, long long ws_floats, void* stream);

/* (ABI 5) The discriminator's pixel-wise head -- FusedLeakyReLU(C) followed by a bias-free 1x1 equalized conv to one plane
 * (reference u_net_2d_discriminator.py:93-97, `final_mapping`; op_static/fused_act.py:64-89 + equalized_layer.py:63-74) -- as one
 * streaming pass per direction over the channels-last bf16 map x [npix][C]:
 *   y[p] = wscale * sum_c w[c] * a[p][c],  a = lrelu(x[p][c] + bias[c]) * scale            (fp32 out; the activated map is not stored)
 *   gx[p][c] = gy[p] * wscale * w[c] * slope(x[p][c] + bias[c]);  grad_bias_and_w[0..C) = sum_p gx,  [C..2C) = wscale * sum_p gy[p] a[p][c]
 * C / 8 a power of two <= 64, bf16 only (MSG_EUNSUPPORTED otherwise).  Sums fp32, overwritten, deterministic
 * (ws_floats >= msg_act_pointwise_head_backward_workspace(npix, C)). */
int msg_act_pointwise_head(const void* x, const float* bias, const float* w, float* y, int dtype,
                           long long npix, int C, float wscale, float alpha, float scale, void* stream);
long long msg_act_pointwise_head_backward_workspace(long long npix, int C);
int msg_act_pointwise_head_backward(const void* x, const float* bias, const float* w, const float* gy, void* gx,
                                    float* grad_bias_and_w, int dtype, long long npix, int C, float wscale,
                                    float alpha, float scale, float* ws, long long ws_floats, void* stream);

/* -------------------------------------------------------------------------
 * Equalized-lr fully connected layers with few rows (mapping network, style affines, classification head), fp32,
 * dense row-major operands.  Replaces F.linear(input, weight * scale, bias * scale_bias) of
 * multi_stylegan/equalized_layer.py (EqualizedLinear.forward) and its autograd derivatives; one launch per call.
 *   msg_linear_fprop   y[M][N]  = gain * x[M][K] . w[N][K]^T + bias_gain * bias[N]   (bias may be NULL)
 *   msg_linear_dgrad   gx[M][K] = gain * gy[M][N] . w[N][K]
 *   msg_linear_wgrad   gw[N][K] = gain * gy[M][N]^T . x[M][K];  gb[N] = bias_gain * sum_m gy[m][n]  (gb may be NULL)
 * ------------------------------------------------------------------------- */
int msg_linear_fprop(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                     float gain, float bias_gain, void* stream);
int msg_linear_dgrad(const float* gy, const float* w, float* gx, int M, int N, int K, float gain, void* stream);
int msg_linear_wgrad(const float* gy, const float* x, float* gw, float* gb, int M, int N, int K,
                     float gain, float bias_gain, void* stream);

/* G layers of the same shape in ONE launch (the generator's style affines, multi_stylegan_generator.py:379-382: every
 * ModulatedConv2d's modulation_mapping applied to its slot of the latent [M][L][K], all known before the first conv):
 * group g reads rows x + slot[g]*K (row pitch L*K), its own weight w[g] ([N][K]) / bias[g] (device pointer tables;
 * `bias` itself may be NULL), and writes the g-th slab of the [G][M][N] (fprop), [G][M][K] (dgrad), [G][N][K] / [G][N]
 * (wgrad) outputs.  Same arithmetic per group as msg_linear_fprop / _dgrad / _wgrad. */
int msg_linear_grouped_fprop(const float* x, const int* slot, const float* const* w, const float* const* bias, float* y,
                             int G, int M, int N, int K, int L, float gain, float bias_gain, void* stream);
int msg_linear_grouped_dgrad(const float* gy, const float* const* w, float* gx, int G, int M, int N, int K, float gain,
                             void* stream);
int msg_linear_grouped_wgrad(const float* gy, const float* x, const int* slot, float* gw, float* gb, int G, int M, int N,
                             int K, int L, float gain, float bias_gain, void* stream);
/* msg_linear_grouped_wgrad with one destination per layer: gw / gb are DEVICE arrays of G device pointers ([N][K] / [N] each;
 * gb may be NULL) -- the layers' slices of a flat gradient store, so that no per-layer accumulation copy follows. */
int msg_linear_grouped_wgrad_ptrs(const float* gy, const float* x, const int* slot, float* const* gw, float* const* gb, int G,
                                  int M, int N, int K, int L, float gain, float bias_gain, void* stream);

/* Which kernel msg_conv2d_fprop launches for a problem (no launch): 5 = the streaming kernels for 1x1 convolutions with
 * <= 8 channels on one side (conv_thin.hip), 3 / 4 = 3x3 row-sharing kernel with the 256x256 / 128x128 tile, 2 = 256x256
 * ping-pong, 1 = 128x128 with LDS-DMA staging, 0 = 128x128 with register staging.  Used by bench.py to label per-kernel timings. */
int msg_conv2d_fprop_plan(int dtype, int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                          int kh, int kw, long long w_batch_stride);
/* 1 if msg_conv2d_fprop takes this problem to the activation-stationary sub-pixel up-convolution kernel (conv_upconv.hip:
 * K = 512, N = 4 * 512, per-sample weights, pixel-shuffled output, bf16), else 0.  For timing labels. */
int msg_conv2d_fprop_upconv_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int kh, int kw,
                                     int stride, int pad, int in_up, int pixel_shuffle, long long w_batch_stride);
/* 1 / 2 if msg_conv2d_fprop takes this bf16 problem to the streaming kernels of conv_thin.hip -- a 1x1 convolution to N <= 8
 * output channels (1: the RGB heads multi_stylegan_generator.py:472-526, the pixel-wise head u_net_2d_discriminator.py:93-97)
 * or from an 8-channel-padded input (2: their data gradients, the first block's residual conv) -- else 0.  act_mode: 0 = plain /
 * bias, 1 = fused activation, 2 = residual merge (as msg_conv2d_fprop_act / _residual).  For timing labels and tests. */
int msg_conv2d_fprop_thin_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy, int kh, int kw,
                                   int stride, int pad, int in_up, int pixel_shuffle, int act_mode);

/* ---------------------------------------------------------------------------
 * Adam over a flat fp32 store, optionally with an exponential moving average of the stepped parameters in the same pass
 * (replaces: torch.optim.Adam.step() at multi_stylegan/model_wrapper.py:296-300, :410-414 and
 * misc.exponential_moving_average, multi_stylegan/misc.py -- ~400 small tensors per network each):
 *   g = grad * coef[0] (coef: device scalar or NULL = 1);  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g^2;
 *   param -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps);  ema = d ema + (1 - d) param (ema != NULL)
 * All arrays n floats, 16-byte aligned; step >= 1 is the number of this update.  msg_flat_ema: the EMA alone.
 * ------------------------------------------------------------------------- */
int msg_flat_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, long long n,
                  const float* coef, float lr, float beta1, float beta2, float eps, int step, float ema_decay,
                  void* stream);
int msg_flat_ema(float* ema, const float* param, long long n, float decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSG_HIP_H */
