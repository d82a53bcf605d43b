// a1 (fast path): the 4x4 FIR blur of the path as a SEPARABLE sliding-window filter, channels-last.
//
// Every FIR the models pass to upfirdn2d is an outer product (k k^T / sum, k = [1,3,3,1]:
// multi_stylegan_generator.py:244-258, u_net_2d_discriminator.py:186-203).  For up = down = 1 (the blur behind every
// upsampling conv of G and every strided conv of D -- the most frequent FIR launch of a training step) the generic
// kernel (upfirdn2d.hip) spends 16 taps per output and keeps a 5x5 input footprint per lane in registers (200 VGPRs,
// two waves per SIMD: it is latency-bound at ~36 % of HBM).  Here a lane owns 16 bytes of channels of TWO adjacent
// output columns and walks down TH rows: per row it loads 5 input vectors, reduces them horizontally (4 taps) to two
// row sums, and the output is the 4-tap vertical combination of the last four row sums, which live in registers
// (a sliding window; the unrolled loop renames them, no moves).  2.5 loads and 64 FMAs per output vector instead of
// 6.25 and 128, ~100 VGPRs (four waves per SIMD), every load a whole 16-B-per-lane row segment.
//
// Result: the same sum in a different association ((x*fx) first, then *fy) -- fp32 rounding-level differences from
// the 2-D form (the host only takes this path when fir == fy fx^T to within 1e-6 relative).
#include "msg_common.h"
#include <stdlib.h>

#ifndef BLUR_ACT_ROUND
// 0 (default): the activation sees the fp32 blur result -- one rounding less than the two-pass form (more accurate), and
// ~17 % faster because the kernel is VALU-bound (127 vs 153 us on the 512-channel maps).  1: round the blur result to the
// storage type first, which reproduces the two-pass form bit for bit (build with -DBLUR_ACT_ROUND=1 to compare).
#define BLUR_ACT_ROUND 0
#endif

struct BlurParams {
    int B, IH, IW, OH, OW, CV;      // CV = 16-byte vectors per pixel (channel stride / VEC)
    int px0, py0;
    int xcd;                        // 1: the workgroups of a row strip are handed to the XCDs in contiguous bands (see the kernel)
    ActEpilogue act;                // optional fused (noise +) bias + leaky ReLU behind the blur (see msg_common.h)
};

// horizontal pass of input row iy -> two row sums (columns ox, ox+1), fp32.  The five vectors come through a buffer
// descriptor over the sample's map: a column outside the image carries an out-of-range offset and the hardware returns
// zeros -- no per-vector select and no zero fill in front of a conditional load (ISA of the previous form: 32 v_cndmask and
// ~20 v_mov per row, a seventh of the kernel's vector instructions, in a kernel that is bound by them).
constexpr int BLUR_OOB = (int)0x80000000;
template <typename T>
__device__ __forceinline__ void blur_hrow(__amdgpu_buffer_rsrc_t rs, int iy, int IH, int row_bytes, int o0, int o1, int o2,
                                          int o3, int o4, f32x4 wx, float (&h)[2][Vec16<T>::N]) {
    constexpr int VEC = Vec16<T>::N;
    // a row outside the image (top / bottom padding; workgroup-uniform) takes out-of-range offsets as well: the row sums come
    // out as zeros without a second code path (an if / else here left part of the caller's window arrays in scratch memory)
    const bool row_ok = (unsigned)iy < (unsigned)IH;
    const int so = row_ok ? iy * row_bytes : 0;
    Vec16<T> v0, v1, v2, v3, v4;
#if defined(__HIP_DEVICE_COMPILE__)
    u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? o0 : BLUR_OOB, so, 0),
          r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? o1 : BLUR_OOB, so, 0),
          r2 = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? o2 : BLUR_OOB, so, 0),
          r3 = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? o3 : BLUR_OOB, so, 0),
          r4 = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? o4 : BLUR_OOB, so, 0);
#else
    u32x4 r0{}, r1{}, r2{}, r3{}, r4{};
#endif
    v0.raw = make_uint4(r0[0], r0[1], r0[2], r0[3]); v1.raw = make_uint4(r1[0], r1[1], r1[2], r1[3]);
    v2.raw = make_uint4(r2[0], r2[1], r2[2], r2[3]); v3.raw = make_uint4(r3[0], r3[1], r3[2], r3[3]);
    v4.raw = make_uint4(r4[0], r4[1], r4[2], r4[3]);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float a0 = v0.get(e), a1 = v1.get(e), a2 = v2.get(e), a3 = v3.get(e), a4 = v4.get(e);
        h[0][e] = fmaf(wx[3], a3, fmaf(wx[2], a2, fmaf(wx[1], a1, wx[0] * a0)));
        h[1][e] = fmaf(wx[3], a4, fmaf(wx[2], a3, fmaf(wx[1], a2, wx[0] * a1)));
    }
}

template <typename T, int TH, int OCC, bool ACT>     // ACT: with the fused noise + bias + leaky-ReLU stage
__global__ __launch_bounds__(256, OCC) void blur_sep_kernel(const T* __restrict__ x, const float* __restrict__ fy,
                                                          const float* __restrict__ fx, T* __restrict__ y,
                                                          BlurParams p) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    // Workgroups are dispatched round-robin over the 8 XCDs in blockIdx.x order, and a workgroup re-reads the three columns
    // to the right of its own: with the strip's workgroups dealt out in x order every such halo column came from another
    // XCD's L2, i.e. from HBM again (PMC: 1.3x the algorithmic traffic).  Remapped, XCD k takes a contiguous band of the strip.
    const unsigned bx = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const long long t = (long long)bx * 256 + threadIdx.x;
    const int xp = (int)(t / p.CV), cv = (int)(t - (long long)xp * p.CV);
    const int ox = 2 * xp;
    if (ox >= p.OW) return;
    const int oy0 = blockIdx.y * TH, b = blockIdx.z;
    // true convolution: tap j of the window meets fir[K-1-j] (upfirdn2d_kernel.cu:100-121 flips the same way)
    f32x4 wx, wy;
    wx[0] = fx[3]; wx[1] = fx[2]; wx[2] = fx[1]; wx[3] = fx[0];
    wy[0] = fy[3]; wy[1] = fy[2]; wy[2] = fy[1]; wy[3] = fy[0];
    // one descriptor per sample (the host checks that a sample's map is below 2 GiB); per-lane byte offsets of the window's
    // five columns (out of range for columns outside the image), the row as the scalar offset
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const uint4*>(x) + (long long)b * p.IH * p.IW * p.CV), 0, 0x7ffffff0, 0x00020000);
    uint4* yout = reinterpret_cast<uint4*>(y) + (long long)b * p.OH * p.OW * p.CV + cv;
    const int c0 = ox - p.px0;
    const int row_bytes = p.IW * p.CV * 16;
    const int o0 = (unsigned)(c0 + 0) < (unsigned)p.IW ? ((c0 + 0) * p.CV + cv) * 16 : BLUR_OOB,
              o1 = (unsigned)(c0 + 1) < (unsigned)p.IW ? ((c0 + 1) * p.CV + cv) * 16 : BLUR_OOB,
              o2 = (unsigned)(c0 + 2) < (unsigned)p.IW ? ((c0 + 2) * p.CV + cv) * 16 : BLUR_OOB,
              o3 = (unsigned)(c0 + 3) < (unsigned)p.IW ? ((c0 + 3) * p.CV + cv) * 16 : BLUR_OOB,
              o4 = (unsigned)(c0 + 4) < (unsigned)p.IW ? ((c0 + 4) * p.CV + cv) * 16 : BLUR_OOB;
    const bool c1ok = ox + 1 < p.OW;

    // fused activation: this lane's bias vector, as ext-vector registers (a float array that is conditionally filled ended up
    // in scratch memory: 48 bytes per lane in the bf16 variant)
    f32x4 a_bias_lo = {0.f, 0.f, 0.f, 0.f}, a_bias_hi = {0.f, 0.f, 0.f, 0.f};
    float a_nw = 0.f;
    if constexpr (ACT) {
        if (p.act.bias) {
            a_bias_lo = *reinterpret_cast<const f32x4*>(p.act.bias + cv * VEC);
            if constexpr (VEC == 8) a_bias_hi = *reinterpret_cast<const f32x4*>(p.act.bias + cv * VEC + 4);
        }
        a_nw = p.act.noise ? p.act.noise_w[0] : 0.f;
    }
    // noise values of the two columns, fetched ONE ROW AHEAD of their use (a dependent load in front of every row's
    // activation would stall the lane once per row)
    const float* nz_base = (ACT && p.act.noise) ? p.act.noise + (long long)(p.act.noise_batch == 1 ? 0 : b) * p.OH * p.OW + ox : nullptr;
    float nz_cur0 = 0.f, nz_cur1 = 0.f;
    if (nz_base && oy0 < p.OH) {
        nz_cur0 = nz_base[(long long)oy0 * p.OW];
        nz_cur1 = (ox + 1 < p.OW) ? nz_base[(long long)oy0 * p.OW + 1] : 0.f;
    }
    float w0[2][VEC], w1[2][VEC], w2[2][VEC];                        // the three previous row sums
    blur_hrow<T>(rs, oy0 - p.py0 + 0, p.IH, row_bytes, o0, o1, o2, o3, o4, wx, w0);
    blur_hrow<T>(rs, oy0 - p.py0 + 1, p.IH, row_bytes, o0, o1, o2, o3, o4, wx, w1);
    blur_hrow<T>(rs, oy0 - p.py0 + 2, p.IH, row_bytes, o0, o1, o2, o3, o4, wx, w2);
#pragma unroll 4
    for (int r = 0; r < TH; ++r) {
        const int oy = oy0 + r;
        if (oy >= p.OH) break;                                       // (uniform)
        float nz_next0 = 0.f, nz_next1 = 0.f;
        if (nz_base && oy + 1 < p.OH) {
            nz_next0 = nz_base[(long long)(oy + 1) * p.OW];
            nz_next1 = c1ok ? nz_base[(long long)(oy + 1) * p.OW + 1] : 0.f;
        }
        float w3[2][VEC];
        blur_hrow<T>(rs, oy - p.py0 + 3, p.IH, row_bytes, o0, o1, o2, o3, o4, wx, w3);
        uint4* dst = yout + ((long long)oy * p.OW + ox) * p.CV;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c == 1 && !c1ok) break;
            float f[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                f[e] = fmaf(wy[3], w3[c][e], fmaf(wy[2], w2[c][e], fmaf(wy[1], w1[c][e], wy[0] * w0[c][e])));
            if constexpr (ACT) {
                // activation of the StyledConv2d that owns this blur (see BLUR_ACT_ROUND above)
                const float nv = a_nw * (c == 0 ? nz_cur0 : nz_cur1);
                const float pos = p.act.scale, neg = p.act.alpha * p.act.scale;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float r;
                    if constexpr (VEC == 4 || !BLUR_ACT_ROUND) r = f[e];
                    else r = bf2f(f2bf(f[e]));
                    const float val = r + (nv + (e < 4 ? a_bias_lo[e & 3] : a_bias_hi[e & 3]));
                    f[e] = BLUR_ACT_ROUND ? ((val > 0.f) ? val : val * p.act.alpha) * p.act.scale : val * ((val > 0.f) ? pos : neg);
                }
            }
            // (packed into four scalars, not through Vec16's element pointer: indexing `(&raw.x)[e]` in this loop made the
            //  compiler build the vector in SCRATCH memory and read it back -- 36 bytes of private segment per lane and
            //  six scratch round trips per output vector in every variant of the kernel)
            u32x4 pk;
            if constexpr (VEC == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = __float_as_uint(f[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (uint32_t)f2bf(f[2 * e]) | ((uint32_t)f2bf(f[2 * e + 1]) << 16);
            }
            *reinterpret_cast<u32x4*>(dst + c * p.CV) = pk;
            if constexpr (ACT && VEC == 8) {
                if (p.act.mask)
                    p.act.mask[(((long long)b * p.OH + oy) * p.OW + ox + c) * p.CV + cv] = (unsigned char)act_sign_byte(pk);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { w0[c][e] = w1[c][e]; w1[c][e] = w2[c][e]; w2[c][e] = w3[c][e]; }
        nz_cur0 = nz_next0;
        nz_cur1 = nz_next1;
    }
}

// x [B, in_h, in_w, minor] channels-last, fir_y [4], fir_x [4] fp32 with fir2d = fir_y fir_x^T; up = down = 1.
static int blur_sep_launch(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                           int major, int in_h, int in_w, int minor, int kh, int kw,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1, const ActEpilogue& act, void* stream);

extern "C" int msg_upfirdn2d_separable(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                                       int major, int in_h, int in_w, int minor, int kh, int kw,
                                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
    ActEpilogue act{};
    return blur_sep_launch(x, fir_y, fir_x, y, dtype, major, in_h, in_w, minor, kh, kw, pad_x0, pad_x1, pad_y0, pad_y1, act,
                           stream);
}

extern "C" int msg_upfirdn2d_separable_act_mask(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                                                int major, int in_h, int in_w, int minor, int kh, int kw,
                                                int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                                const float* act_bias, const float* noise, const float* noise_weight,
                                                int noise_batch, float alpha, float scale, unsigned char* mask, void* stream) {
    if (noise && (!noise_weight || (noise_batch != 1 && noise_batch != major))) return MSG_EINVAL;
    if (mask && (dtype != MSG_BF16 || minor % 8)) return MSG_EUNSUPPORTED;
    ActEpilogue act{act_bias, noise, noise_weight, noise_batch, 1, alpha, scale, nullptr, 0, 0.f, mask};
    return blur_sep_launch(x, fir_y, fir_x, y, dtype, major, in_h, in_w, minor, kh, kw, pad_x0, pad_x1, pad_y0, pad_y1, act,
                           stream);
}

extern "C" int msg_upfirdn2d_separable_act(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                                           int major, int in_h, int in_w, int minor, int kh, int kw,
                                           int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                           const float* act_bias, const float* noise, const float* noise_weight,
                                           int noise_batch, float alpha, float scale, void* stream) {
    return msg_upfirdn2d_separable_act_mask(x, fir_y, fir_x, y, dtype, major, in_h, in_w, minor, kh, kw, pad_x0, pad_x1, pad_y0,
                                            pad_y1, act_bias, noise, noise_weight, noise_batch, alpha, scale, nullptr, stream);
}

static int blur_sep_launch(const void* x, const float* fir_y, const float* fir_x, void* y, int dtype,
                           int major, int in_h, int in_w, int minor, int kh, int kw,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1, const ActEpilogue& act, void* stream) {
    if (major == 0) return MSG_OK;
    if (!x || !fir_y || !fir_x || !y || major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (kh != 4 || kw != 4 || minor % vec || (((uintptr_t)x | (uintptr_t)y) & 15u)) return MSG_EUNSUPPORTED;
    const int oh = in_h + pad_y0 + pad_y1 - kh + 1, ow = in_w + pad_x0 + pad_x1 - kw + 1;
    if (oh <= 0 || ow <= 0) return MSG_EINVAL;
    if ((long long)in_h * in_w * minor * (dtype == MSG_BF16 ? 2 : 4) >= 0x7ffffff0ll) return MSG_EUNSUPPORTED;   // 31-bit buffer offsets per sample
    static const int xcd = msg_tunable("MSG_BLUR_XCD", 1);
    BlurParams p{major, in_h, in_w, oh, ow, minor / vec, pad_x0, pad_y0, 0, act};
    static const int variant = msg_tunable("MSG_BLUR_VARIANT", 0);
    const long long threads = (long long)((ow + 1) / 2) * p.CV;
    const long long gx = (threads + 255) / 256;
    // rows per lane: 32 amortises the 3 warm-up rows best (4.36 vs 3.48 TB/s on 512ch @256^2), as long as the grid still
    // has >= 1024 workgroups; smaller maps take 16-row strips (31 vs 40 us on 128ch @128^2)
    int th = (gx * ((oh + 31) / 32) * major >= 1024) ? 32 : 16;
    if (variant == 1) th = 32;
    if (variant == 2) th = 16;
    const int gy = (oh + th - 1) / th;
    if (gx >= (1ll << 31) || gy > 65535 || major > 65535) return MSG_EUNSUPPORTED;
    dim3 grid((unsigned)gx, gy, major);
    p.xcd = xcd && gx >= 16 && gx % 8 == 0;      // (XCD of a workgroup = linear id % 8 = blockIdx.x % 8 only then)
    hipStream_t s = (hipStream_t)stream;
#define BLUR_LAUNCH(TH_, ACT_, OCC_)                                                                                      \
    do {                                                                                                               \
        if (dtype == MSG_BF16)                                                                                         \
            hipLaunchKernelGGL((blur_sep_kernel<bf16_t, TH_, OCC_, ACT_>), grid, dim3(256), 0, s, (const bf16_t*)x,   \
                               fir_y, fir_x, (bf16_t*)y, p);                                                          \
        else                                                                                                           \
            hipLaunchKernelGGL((blur_sep_kernel<float, TH_, 4, ACT_>), grid, dim3(256), 0, s, (const float*)x, fir_y, \
                               fir_x, (float*)y, p);                                                                  \
    } while (0)
    // the bf16 kernel with the activation stage needs ~140 registers: at four waves per SIMD (128) it spills its window,
    // at three it does not
    static const int act_occ = msg_tunable("MSG_BLUR_ACT_OCC", 3);
    if (act.enabled) {
        if (act_occ == 3) { if (th == 32) BLUR_LAUNCH(32, true, 3); else BLUR_LAUNCH(16, true, 3); }
        else { if (th == 32) BLUR_LAUNCH(32, true, 4); else BLUR_LAUNCH(16, true, 4); }
    } else {
        if (th == 32) BLUR_LAUNCH(32, false, 4);
        else BLUR_LAUNCH(16, false, 4);
    }
#undef BLUR_LAUNCH
    return MSG_CHECK_LAUNCH();
}
