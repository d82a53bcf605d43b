// a3: weight modulation / demodulation of the dual-styled convolution (multi_stylegan_generator.py:379-388) as
// three small gfx950 kernels, so that no [B,O,I,kh,kw] fp32 tensor is ever materialised by elementwise library ops:
//
//   msg_demod_coeff        d[b,o] = rsqrt(scale^2 * sum_{i,t} (W[o,i,t] * s[b,i])^2 + eps)
//                          one workgroup per output channel, wave-shuffle + LDS reduction per sample
//   msg_scale_rows_cols    out[b][r][t][c] = base[r][t][c] * rowscale[b][r] * colscale[b][c]   (c >= C: 0)
//                          writes the per-sample weights straight in the contraction kernels' K-contiguous layouts
//                          (forward: r = o, c = i;  data gradient: r = i, c = o), in bf16 or f32
//   msg_modulate_backward  from the per-sample weight gradients GWK[b][o][t][i] (what conv_wgrad produces) to
//                          gW[o][i][t] (summed over the batch) and the style gradient gs[b][i], including the
//                          derivative of the demodulation norm:  with u = scale*W*s, w = d*u,
//                          g_u = d*g_w - d^3 * u * <g_w, u>_{i,t}
#include "msg_common.h"
#include <stdlib.h>

__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// W [O][I][T] fp32, s [B][I] fp32 -> d [B][O] fp32
__global__ __launch_bounds__(256) void demod_coeff_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                          float* __restrict__ d, int B, int O, int I, int T,
                                                          float scale, float eps) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const float* wo = W + (size_t)o * I * T;
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < I; i += 256) {
            const float sv = s[(size_t)b * I + i];
            float w2 = 0.f;
            for (int t = 0; t < T; ++t) { const float wv = wo[i * T + t]; w2 = fmaf(wv, wv, w2); }
            acc = fmaf(w2, sv * sv, acc);
        }
        const float tot = block_sum_256(acc, red);
        if (threadIdx.x == 0) d[(size_t)b * O + o] = rsqrtf(scale * scale * tot + eps);
    }
}

extern "C" int msg_demod_coeff(const float* W, const float* s, float* d, int B, int O, int I, int taps,
                               float scale, float eps, void* stream) {
    if (B == 0) return MSG_OK;
    if (!W || !s || !d || B < 0 || O <= 0 || I <= 0 || taps <= 0) return MSG_EINVAL;
    hipLaunchKernelGGL(demod_coeff_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, W, s, d, B, O, I, taps, scale, eps);
    return MSG_CHECK_LAUNCH();
}

// VEC consecutive floats as float4 loads when the run is whole and 16-B aligned, element-wise (zero past `limit`) otherwise
template <int VEC>
__device__ __forceinline__ void load_run(const float* __restrict__ src, int c0, int limit, bool aligned, float (&f)[VEC]) {
    if (aligned && c0 + VEC <= limit) {
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + c0 + e);
            f[e] = v.x; f[e + 1] = v.y; f[e + 2] = v.z; f[e + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = (c0 + e < limit) ? src[c0 + e] : 0.f;
    }
}

// base [R][T][C] fp32; rowscale [B][R] (or NULL = 1); colscale [B][C] (or NULL = 1); out [B][R][T][Ck] of type TO.
// grid = (row r, group of BG samples): a thread owns up to ITEMS (tap, 16-B column vector) cells of the row, reads
// their fp32 base values ONCE into registers and then emits the BG per-sample copies (row scale = one scalar per
// sample, column scales = one vector per lane and sample); no index is ever divided.  The base row is therefore read
// B/BG times instead of B times and every workgroup streams BG * T * Ck elements.
constexpr int SRC_ITEMS = 5;
template <typename TO>
__global__ __launch_bounds__(256) void scale_rows_cols_kernel(const float* __restrict__ base,
                                                              const float* __restrict__ rowscale,
                                                              const float* __restrict__ colscale, TO* __restrict__ out,
                                                              int B, int BG, int R, int T, int C, int Ck, float gain) {
    using V = Vec16<TO>;
    constexpr int VEC = V::N;
    const int r = blockIdx.x, b0 = blockIdx.y * BG, b1 = min(B, b0 + BG);
    const int cvecs = Ck / VEC;
    const float* src_row = base + (size_t)r * T * C;
    const int tstep = 256 / cvecs > 0 ? 256 / cvecs : 1;
    const bool al = (C % 4 == 0) && ((((uintptr_t)base | (uintptr_t)colscale) & 15u) == 0);   // float4-loadable runs
    for (int cv = threadIdx.x % cvecs; cv < cvecs; cv += 256) {       // (cvecs <= 256 in practice: one pass)
        const int c0 = cv * VEC;
        const int t0 = threadIdx.x / cvecs;
        for (int tb = t0; tb < T; tb += tstep * SRC_ITEMS) {           // (T <= tstep * SRC_ITEMS in practice: one pass)
            float f[SRC_ITEMS][VEC];
#pragma unroll
            for (int k = 0; k < SRC_ITEMS; ++k) {
                const int t = tb + k * tstep;
                if (t < T) load_run<VEC>(src_row + (size_t)t * C, c0, C, al, f[k]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[k][e] = 0.f;
                }
            }
            for (int b = b0; b < b1; ++b) {
                const float rs = gain * (rowscale ? rowscale[(size_t)b * R + r] : 1.f);
                float sc[VEC];
                if (colscale) load_run<VEC>(colscale + (size_t)b * C, c0, C, al, sc);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) sc[e] = (c0 + e < C) ? 1.f : 0.f;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) sc[e] *= rs;
                TO* dst_row = out + ((size_t)b * R + r) * T * Ck;
#pragma unroll
                for (int k = 0; k < SRC_ITEMS; ++k) {
                    const int t = tb + k * tstep;
                    if (t >= T) break;
                    V o;
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.set(e, f[k][e] * sc[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.set2(e, f[k][2 * e] * sc[2 * e], f[k][2 * e + 1] * sc[2 * e + 1]);
                    }
                    *reinterpret_cast<uint4*>(dst_row + (size_t)t * Ck + c0) = o.raw;
                }
            }
        }
    }
}

extern "C" int msg_scale_rows_cols(const float* base, const float* rowscale, const float* colscale, void* out,
                                   int dtype, int B, int R, int T, int C, int Ck, float gain, void* stream) {
    if (B == 0) return MSG_OK;
    if (!base || !out || B < 0 || R <= 0 || T <= 0 || C <= 0 || Ck < C) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (Ck % vec || ((uintptr_t)out & 15u) || B > 65535) return MSG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    // samples per workgroup: as many as keep >= ~2048 workgroups in flight (256 CUs x 8), at most 8
    int bg = 1;
    while (bg < 8 && bg * 2 <= B && (long long)R * ((B + 2 * bg - 1) / (2 * bg)) >= 2048) bg *= 2;
    dim3 grid(R, (B + bg - 1) / bg);
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((scale_rows_cols_kernel<bf16_t>), grid, dim3(256), 0, s, base, rowscale, colscale,
                           (bf16_t*)out, B, bg, R, T, C, Ck, gain);
    else
        hipLaunchKernelGGL((scale_rows_cols_kernel<float>), grid, dim3(256), 0, s, base, rowscale, colscale,
                           (float*)out, B, bg, R, T, C, Ck, gain);
    return MSG_CHECK_LAUNCH();
}

// out[b][r][t][c] = gain * base[r][t][c] * (row1[b][r] * col1[b][c] + row2[b][r] * col2[b][c]),  c >= C -> 0.
// The directional derivative of the per-sample weight set w = scale * d * W * s along a style direction v:
//   dw = scale * W * (d (x) v + dd (x) s),   dd[b][o] = -scale^2 d^3 sum_i s v wsq[o][i]
// (row1 = d, col1 = v, row2 = dd, col2 = s for the forward image; rows and columns swapped for the data-gradient image)
// written straight in the contraction kernels' layouts -- the weights of the second-order contractions F(x, dw), D(gy, dw)
// of the path-length pass.  Same tiling as scale_rows_cols_kernel.
template <typename TO>
__global__ __launch_bounds__(256) void scale_rows_cols2_kernel(const float* __restrict__ base,
                                                               const float* __restrict__ row1, const float* __restrict__ col1,
                                                               const float* __restrict__ row2, const float* __restrict__ col2,
                                                               TO* __restrict__ out, int B, int BG, int R, int T, int C, int Ck,
                                                               float gain) {
    using V = Vec16<TO>;
    constexpr int VEC = V::N;
    const int r = blockIdx.x, b0 = blockIdx.y * BG, b1 = min(B, b0 + BG);
    const int cvecs = Ck / VEC;
    const float* src_row = base + (size_t)r * T * C;
    const int tstep = 256 / cvecs > 0 ? 256 / cvecs : 1;
    const bool al = (C % 4 == 0) && ((((uintptr_t)base | (uintptr_t)col1 | (uintptr_t)col2) & 15u) == 0);
    for (int cv = threadIdx.x % cvecs; cv < cvecs; cv += 256) {
        const int c0 = cv * VEC;
        const int t0 = threadIdx.x / cvecs;
        for (int tb = t0; tb < T; tb += tstep * SRC_ITEMS) {
            float f[SRC_ITEMS][VEC];
#pragma unroll
            for (int k = 0; k < SRC_ITEMS; ++k) {
                const int t = tb + k * tstep;
                if (t < T) load_run<VEC>(src_row + (size_t)t * C, c0, C, al, f[k]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[k][e] = 0.f;
                }
            }
            for (int b = b0; b < b1; ++b) {
                const float rs1 = gain * row1[(size_t)b * R + r], rs2 = gain * row2[(size_t)b * R + r];
                float sc[VEC], s2[VEC];
                load_run<VEC>(col1 + (size_t)b * C, c0, C, al, sc);
                load_run<VEC>(col2 + (size_t)b * C, c0, C, al, s2);
#pragma unroll
                for (int e = 0; e < VEC; ++e) sc[e] = fmaf(rs1, sc[e], rs2 * s2[e]);
                TO* dst_row = out + ((size_t)b * R + r) * T * Ck;
#pragma unroll
                for (int k = 0; k < SRC_ITEMS; ++k) {
                    const int t = tb + k * tstep;
                    if (t >= T) break;
                    V o;
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.set(e, f[k][e] * sc[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.set2(e, f[k][2 * e] * sc[2 * e], f[k][2 * e + 1] * sc[2 * e + 1]);
                    }
                    *reinterpret_cast<uint4*>(dst_row + (size_t)t * Ck + c0) = o.raw;
                }
            }
        }
    }
}

extern "C" int msg_scale_rows_cols2(const float* base, const float* row1, const float* col1, const float* row2,
                                    const float* col2, void* out, int dtype, int B, int R, int T, int C, int Ck, float gain,
                                    void* stream) {
    if (B == 0) return MSG_OK;
    if (!base || !row1 || !col1 || !row2 || !col2 || !out || B < 0 || R <= 0 || T <= 0 || C <= 0 || Ck < C) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (Ck % vec || (((uintptr_t)out) & 15u)) return MSG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int bg = 1;
    while (bg < 8 && bg * 2 <= B && (long long)R * ((B + 2 * bg - 1) / (2 * bg)) >= 2048) bg *= 2;
    dim3 grid(R, (B + bg - 1) / bg);
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((scale_rows_cols2_kernel<bf16_t>), grid, dim3(256), 0, s, base, row1, col1, row2, col2,
                           (bf16_t*)out, B, bg, R, T, C, Ck, gain);
    else
        hipLaunchKernelGGL((scale_rows_cols2_kernel<float>), grid, dim3(256), 0, s, base, row1, col1, row2, col2,
                           (float*)out, B, bg, R, T, C, Ck, gain);
    return MSG_CHECK_LAUNCH();
}

// Forward weight set of the modulated conv in ONE launch: demodulation coefficient + per-sample weights.
//   d[b][o]        = rsqrt(scale^2 * sum_i s[b][i]^2 * wsq[o][i] + eps)        (wsq[o][i] = sum_t W[o][i][t]^2, cached)
//   out[b][r][t][c] = scale * d[b][r % O] * base[r][t][c] * s[b][c]
// Same tiling as scale_rows_cols (row r, group of BG samples); the BG coefficients of the row come from one block
// reduction over the input channels.  Replaces msg_demod_coeff (which re-read all of W per call) + msg_scale_rows_cols.
// A workgroup takes ALL rows that share its output channel -- r = o, o + O, ... (the four sub-pixel rows of the 2x2
// transposed conv, R = 4 O with one tap each): one reduction for them instead of four, and 4 x BG KiB per workgroup instead
// of BG (round 4, tools/modw_probe.py: the up-conv's weight set 29.6 us for 33.6 MB = 1.3 TB/s with one 1-KiB row per
// workgroup and sample).
template <typename TO>
__global__ __launch_bounds__(256) void modulate_weights_kernel(const float* __restrict__ base, const float* __restrict__ wsq,
                                                               const float* __restrict__ style, TO* __restrict__ out,
                                                               float* __restrict__ d_out, int B, int BG, int R, int O,
                                                               int T, int C, int Ck, float scale, float eps) {
    using V = Vec16<TO>;
    constexpr int VEC = V::N;
    __shared__ float red[4 * 8];
    __shared__ float dsh[8];
    const int o = blockIdx.x, b0 = blockIdx.y * BG, b1 = min(B, b0 + BG);
    // ---- demodulation coefficients of this row for the BG samples
    {
        float part[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) part[k] = 0.f;
        for (int i = threadIdx.x; i < C; i += 256) {
            const float q = wsq[(size_t)o * C + i];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (b0 + k < b1) { const float sv = style[(size_t)(b0 + k) * C + i]; part[k] = fmaf(q, sv * sv, part[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = wave_sum(part[k]);
            if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 8 + k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 8 && b0 + (int)threadIdx.x < b1) {
            const int k = threadIdx.x;
            const float tot = red[k] + red[8 + k] + red[16 + k] + red[24 + k];
            const float dv = rsqrtf(scale * scale * tot + eps);
            dsh[k] = dv;
            if (d_out) d_out[(size_t)(b0 + k) * O + o] = dv;
        }
        __syncthreads();
    }
    const int cvecs = Ck / VEC;
    const int tstep = 256 / cvecs > 0 ? 256 / cvecs : 1;
    const bool al = (C % 4 == 0) && ((((uintptr_t)base | (uintptr_t)style) & 15u) == 0);   // float4-loadable runs
    // cells of this workgroup: (row, tap) pairs q = (r - o) / O * T + t of the rows that share output channel o, times the
    // row's 16-byte column vectors; a thread keeps one column vector (its style values are loaded once per sample) and walks
    // the pairs -- with one tap per row (the up-conv) the four sub-pixel rows fill the 256 threads that a single row left 3/4 idle
    const int npairs = (R / O) * T;
    for (int cv = threadIdx.x % cvecs; cv < cvecs; cv += 256) {
        const int c0 = cv * VEC;
        const int t0 = threadIdx.x / cvecs;
        for (int qb = t0; qb < npairs; qb += tstep * SRC_ITEMS) {
            float f[SRC_ITEMS][VEC];
            size_t cell[SRC_ITEMS];                                   // (row * T + tap) of item k
#pragma unroll
            for (int k = 0; k < SRC_ITEMS; ++k) {
                const int q = qb + k * tstep;
                const int rk = q / T, t = q - rk * T;
                cell[k] = (size_t)(o + rk * O) * T + t;
                if (q < npairs) load_run<VEC>(base + cell[k] * C, c0, C, al, f[k]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[k][e] = 0.f;
                }
            }
            for (int b = b0; b < b1; ++b) {
                const float rs = scale * dsh[b - b0];
                float sc[VEC];
                load_run<VEC>(style + (size_t)b * C, c0, C, al, sc);
#pragma unroll
                for (int e = 0; e < VEC; ++e) sc[e] *= rs;
                TO* dst_b = out + (size_t)b * R * T * Ck;
#pragma unroll
                for (int k = 0; k < SRC_ITEMS; ++k) {
                    if (qb + k * tstep >= npairs) break;
                    V ov;
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov.set(e, f[k][e] * sc[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov.set2(e, f[k][2 * e] * sc[2 * e], f[k][2 * e + 1] * sc[2 * e + 1]);
                    }
                    *reinterpret_cast<uint4*>(dst_b + cell[k] * Ck + c0) = ov.raw;
                }
            }
        }
    }
}

extern "C" int msg_modulate_weights(const float* base, const float* wsq, const float* style, void* out, float* d_out,
                                    int dtype, int B, int R, int O, int T, int C, int Ck, float scale, float eps,
                                    void* stream) {
    if (B == 0) return MSG_OK;
    if (!base || !wsq || !style || !out || B < 0 || R <= 0 || O <= 0 || R % O || T <= 0 || C <= 0 || Ck < C) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (Ck % vec || ((uintptr_t)out & 15u) || B > 65535) return MSG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    // samples per workgroup double while the grid keeps this many workgroups (round 4: 1024, not 2048 -- 25.4 -> 22.7 us on
    // 512 x 9 x 512 at batch 16: fewer, larger workgroups amortise the reduction's two barriers better)
    static const int min_wgs = msg_tunable("MSG_MODW_MIN_WGS", 1024);
    int bg = 1;
    while (bg < 8 && bg * 2 <= B && (long long)O * ((B + 2 * bg - 1) / (2 * bg)) >= min_wgs) bg *= 2;
    dim3 grid(O, (B + bg - 1) / bg);
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((modulate_weights_kernel<bf16_t>), grid, dim3(256), 0, s, base, wsq, style, (bf16_t*)out, d_out,
                           B, bg, R, O, T, C, Ck, scale, eps);
    else
        hipLaunchKernelGGL((modulate_weights_kernel<float>), grid, dim3(256), 0, s, base, wsq, style, (float*)out, d_out,
                           B, bg, R, O, T, C, Ck, scale, eps);
    return MSG_CHECK_LAUNCH();
}

// One workgroup per group of OG output channels.  Thread = input channel(s) i = tid + 256*slot (all taps of it):
// the style gradient of its channels never leaves the thread; <g_w,u> needs one block reduction per (o, all b at once).
constexpr int MB_SLOTS = 2;      // I <= 512
constexpr int MB_TAPS = 9;
constexpr int MB_BMAX = 32;
template <bool DEMOD>
__global__ __launch_bounds__(256) void modulate_backward_kernel(const float* __restrict__ gwk, const float* __restrict__ W,
                                                                const float* __restrict__ s, const float* __restrict__ d,
                                                                float* __restrict__ gW, float* __restrict__ gs_part,
                                                                int B, int O, int I, int T, int ldg, int OG, float scale) {
    __shared__ float red[4 * MB_BMAX];
    __shared__ float Tsh[MB_BMAX];
    const int og = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float gs_acc[MB_BMAX / 2][MB_SLOTS];             // filled for b < B (B <= 16 fits the unrolled part; see host check)
#pragma unroll
    for (int b = 0; b < MB_BMAX / 2; ++b)
#pragma unroll
        for (int k = 0; k < MB_SLOTS; ++k) gs_acc[b][k] = 0.f;

    for (int oo = 0; oo < OG; ++oo) {
        const int o = og * OG + oo;
        if (o >= O) break;
        float wv[MB_SLOTS][MB_TAPS], gacc[MB_SLOTS][MB_TAPS];
#pragma unroll
        for (int k = 0; k < MB_SLOTS; ++k) {
            const int i = tid + 256 * k;
#pragma unroll
            for (int t = 0; t < MB_TAPS; ++t) {
                wv[k][t] = (i < I && t < T) ? W[((size_t)o * I + i) * T + t] : 0.f;
                gacc[k][t] = 0.f;
            }
        }
        if (DEMOD) {
            // T[b] = <g_w[b,o], u[b,o]> for every sample, one reduction round for all of them
            for (int b = 0; b < B; ++b) {
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < MB_SLOTS; ++k) {
                    const int i = tid + 256 * k;
                    if (i >= I) continue;
                    const float sv = s[(size_t)b * I + i];
                    const float* g = gwk + (((size_t)b * O + o) * T) * ldg + i;
#pragma unroll
                    for (int t = 0; t < MB_TAPS; ++t)
                        if (t < T) part = fmaf(g[(size_t)t * ldg], wv[k][t] * sv, part);
                }
                part = wave_sum(part);
                if (lane == 0) red[wid * MB_BMAX + b] = part;
            }
            __syncthreads();
            if (tid < B) Tsh[tid] = scale * (red[tid] + red[MB_BMAX + tid] + red[2 * MB_BMAX + tid] + red[3 * MB_BMAX + tid]);
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < MB_BMAX / 2; ++b) {
            if (b >= B) break;
            const float dv = DEMOD ? d[(size_t)b * O + o] : 1.f;
            const float corr = DEMOD ? dv * dv * dv * Tsh[b] : 0.f;
#pragma unroll
            for (int k = 0; k < MB_SLOTS; ++k) {
                const int i = tid + 256 * k;
                if (i >= I) continue;
                const float sv = s[(size_t)b * I + i];
                const float* g = gwk + (((size_t)b * O + o) * T) * ldg + i;
                float sgrad = 0.f;
#pragma unroll
                for (int t = 0; t < MB_TAPS; ++t) {
                    if (t >= T) continue;
                    // g_u = d * g_w - d^3 * u * <g_w,u>,  u = scale * W * s
                    const float gu = dv * g[(size_t)t * ldg] - corr * (scale * wv[k][t] * sv);
                    gacc[k][t] = fmaf(gu, sv, gacc[k][t]);
                    sgrad = fmaf(gu, wv[k][t], sgrad);
                }
                gs_acc[b][k] += sgrad;
            }
        }
#pragma unroll
        for (int k = 0; k < MB_SLOTS; ++k) {
            const int i = tid + 256 * k;
            if (i >= I) continue;
#pragma unroll
            for (int t = 0; t < MB_TAPS; ++t)
                if (t < T) gW[((size_t)o * I + i) * T + t] = scale * gacc[k][t];
        }
        if (DEMOD) __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < MB_BMAX / 2; ++b) {
        if (b >= B) break;
#pragma unroll
        for (int k = 0; k < MB_SLOTS; ++k) {
            const int i = tid + 256 * k;
            if (i < I) gs_part[((size_t)og * B + b) * I + i] = scale * gs_acc[b][k];
        }
    }
}

// The same fold for I % 4 == 0 and T in {1, 4, 9} (every modulated conv of the generator), restructured around the
// only large operand, the per-sample weight gradients g_w [B][O][T][ldg] (151 MB for a 512x512 3x3 layer at B = 16):
//  * everything the fold needs from g_w is LINEAR in it -- g_u = d g_w - d^3 u <g_w,u> gives
//        gW[o,i,t]  = scale * ( sum_b d_b s_bi g_w  -  scale W_oit sum_b d_b^3 T_b s_bi^2 ),
//        gs[b,i]   += scale * ( d_b sum_t W_oit g_w  -  d_b^3 T_b scale s_bi sum_t W_oit^2 ),   T_b = scale <g_w, W s_b>
//    so g_w is read ONCE (the old kernel read it for T_b and again for the sums), with no load depending on a
//    reduction: all of a thread's loads for an output channel are in flight together;
//  * a thread owns FOUR consecutive input channels: 16-byte loads of g_w, s and W, 16-byte stores of gW, gs;
//  * the two halves of the workgroup take the even / the odd samples and meet in LDS for gW.
template <bool DEMOD, int T>
__global__ __launch_bounds__(256) void modulate_backward_v4_kernel(const float* __restrict__ gwk, const float* __restrict__ W,
                                                                   const float* __restrict__ s, const float* __restrict__ d,
                                                                   float* __restrict__ gW, float* __restrict__ gs_part,
                                                                   int B, int O, int I, int ldg, int OG, float scale) {
    constexpr int NB = MB_BMAX / 4;                   // samples per half (B <= 16)
    __shared__ float red[4][NB];
    __shared__ float Tsh[2][NB];
    __shared__ __attribute__((aligned(16))) float xch[128][4 * T + 4];
    const int og = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = tid >> 7, i = (tid & 127) * 4;
    const bool live = i < I;
    float gs_acc[NB][4];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) gs_acc[q][e] = 0.f;

    for (int oo = 0; oo < OG; ++oo) {
        const int o = og * OG + oo;
        if (o >= O) break;
        float wv[4][T], gl[4][T], p1[NB][4], w2[4], c2[4];
        {
            float flat[4 * T];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const f32x4 v = live ? *reinterpret_cast<const f32x4*>(W + ((size_t)o * I + i) * T + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) flat[4 * j + e] = v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w2[e] = 0.f; c2[e] = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    wv[e][t] = flat[e * T + t];
                    gl[e][t] = 0.f;
                    w2[e] = fmaf(wv[e][t], wv[e][t], w2[e]);
                }
            }
        }
        float part[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int b = 2 * q + half;
            part[q] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) p1[q][e] = 0.f;
            if (b >= B || !live) continue;
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (size_t)b * I + i);
            const float dv = DEMOD ? d[(size_t)b * O + o] : 1.f;
            const float* g = gwk + (((size_t)b * O + o) * T) * ldg + i;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (size_t)t * ldg);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p1[q][e] = fmaf(wv[e][t], gv[e], p1[q][e]);
                    gl[e][t] = fmaf(dv * sv[e], gv[e], gl[e][t]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) part[q] = fmaf(sv[e], p1[q][e], part[q]);
        }
        if (DEMOD) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const float tot = wave_sum(part[q]);
                if (lane == 0) red[wid][q] = tot;
            }
            __syncthreads();
            if (tid < 2 * NB) Tsh[tid / NB][tid % NB] = scale * (red[2 * (tid / NB)][tid % NB] + red[2 * (tid / NB) + 1][tid % NB]);
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int b = 2 * q + half;
            if (b >= B || !live) continue;
            const float dv = DEMOD ? d[(size_t)b * O + o] : 1.f;
            const float corr = DEMOD ? dv * dv * dv * Tsh[half][q] : 0.f;
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (size_t)b * I + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gs_acc[q][e] += dv * p1[q][e] - corr * scale * sv[e] * w2[e];
                c2[e] = fmaf(corr * sv[e], sv[e], c2[e]);
            }
        }
        // gW = scale * (gl - scale * W * c2), both halves added
        if (half == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < T; ++t) xch[tid & 127][e * T + t] = gl[e][t] - scale * wv[e][t] * c2[e];
        }
        __syncthreads();
        if (half == 0 && live) {
            float outv[4 * T];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    outv[e * T + t] = scale * (gl[e][t] - scale * wv[e][t] * c2[e] + xch[tid][e * T + t]);
#pragma unroll
            for (int j = 0; j < T; ++j)
                *reinterpret_cast<f32x4*>(gW + ((size_t)o * I + i) * T + 4 * j) =
                    f32x4{outv[4 * j], outv[4 * j + 1], outv[4 * j + 2], outv[4 * j + 3]};
        }
        __syncthreads();
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int b = 2 * q + half;
            if (b >= B) continue;
            *reinterpret_cast<f32x4*>(gs_part + ((size_t)og * B + b) * I + i) =
                f32x4{scale * gs_acc[q][0], scale * gs_acc[q][1], scale * gs_acc[q][2], scale * gs_acc[q][3]};
        }
    }
}

// Second-order fold of the modulated convolution (path-length regularisation differentiates the style gradient gs of the
// first backward once more).  With w = c d W s (c = scale, d = (c^2 A + eps)^-1/2, A = sum W^2 s^2) the first backward is
//   (gW, gs) = MB(g; W, s),  g = per-sample weight gradient;     for a cotangent v of gs,  L2 = <v, gs> reads
//   L2 = sum_{b,o} [ c d R1 - 1/2 c^3 d^3 dA P ],   Q[b,o,i] = sum_t g W,  R1 = sum_i v Q,  P = sum_i s Q,
//                                                 dA = 2 sum_i s v wsq[o,i]              (wsq = sum_t W^2)
// and this kernel returns its derivatives with respect to W and s at fixed g (the derivative with respect to g is the
// weight set msg_scale_rows_cols2 writes):
//   dL2/dW[o,i,t] = sum_b { g (c d v - 1/2 c^3 d^3 dA s) + W (-c^3 d^3 R1 s^2 - 2 c^3 d^3 P s v + 3/2 c^5 d^5 dA P s^2) }
//   dL2/ds[b,i]   = sum_o { wsq (-c^3 d^3 R1 s + 3/2 c^5 d^5 dA P s - c^3 d^3 P v) - 1/2 c^3 d^3 dA Q }
// Without demodulation d = 1 and every d-derivative vanishes: dL2/dW = c sum_b g v, dL2/ds = 0.
// Same organisation as modulate_backward_v4_kernel (a thread owns four input channels, the halves of the workgroup take the
// even / odd samples, g is read once); dA needs no g and is reduced first, P and R1 together behind the loads.
template <bool DEMOD, int T>
__global__ __launch_bounds__(256) void modulate_backward2_v4_kernel(const float* __restrict__ gwk, const float* __restrict__ W,
                                                                    const float* __restrict__ s, const float* __restrict__ d,
                                                                    const float* __restrict__ v, float* __restrict__ gW,
                                                                    float* __restrict__ gs_part, int B, int O, int I, int ldg,
                                                                    int OG, float scale) {
    constexpr int NB = MB_BMAX / 4;                   // samples per half (B <= 16)
    __shared__ float red[3][4][NB];
    __shared__ float tot[3][2][NB];                   // [dA, P, R1][half][sample]
    __shared__ __attribute__((aligned(16))) float xch[128][4 * T + 4];
    const int og = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = tid >> 7, i = (tid & 127) * 4;
    const bool live = i < I;
    const float c1 = scale, c3 = scale * scale * scale, c5 = c3 * scale * scale;
    float gs_acc[NB][4];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) gs_acc[q][e] = 0.f;

    for (int oo = 0; oo < OG; ++oo) {
        const int o = og * OG + oo;
        if (o >= O) break;
        float wv[4][T], gl[4][T], p1[NB][4], w2[4], c2[4];
        {
            float flat[4 * T];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const f32x4 t4 = live ? *reinterpret_cast<const f32x4*>(W + ((size_t)o * I + i) * T + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) flat[4 * j + e] = t4[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w2[e] = 0.f; c2[e] = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    wv[e][t] = flat[e * T + t];
                    gl[e][t] = 0.f;
                    w2[e] = fmaf(wv[e][t], wv[e][t], w2[e]);
                }
            }
        }
        // ---- dA[b] = 2 sum_i s v wsq  (no g involved)
        if (DEMOD) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int b = 2 * q + half;
                float part = 0.f;
                if (b < B && live) {
                    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (size_t)b * I + i);
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (size_t)b * I + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) part = fmaf(sv[e] * vv[e], w2[e], part);
                }
                part = wave_sum(part);
                if (lane == 0) red[0][wid][q] = part;
            }
            __syncthreads();
            if (tid < 2 * NB) tot[0][tid / NB][tid % NB] = 2.f * (red[0][2 * (tid / NB)][tid % NB] + red[0][2 * (tid / NB) + 1][tid % NB]);
            __syncthreads();
        }
        // ---- one pass over g: Q, the partial sums of P and R1, and the g-term of dL2/dW
        float partP[NB], partR[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int b = 2 * q + half;
            partP[q] = 0.f; partR[q] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) p1[q][e] = 0.f;
            if (b >= B || !live) continue;
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (size_t)b * I + i);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (size_t)b * I + i);
            const float dv = DEMOD ? d[(size_t)b * O + o] : 1.f;
            const float rho1 = c1 * dv;
            const float rho2 = DEMOD ? -0.5f * c3 * dv * dv * dv * tot[0][half][q] : 0.f;
            float coef[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) coef[e] = fmaf(rho1, vv[e], rho2 * sv[e]);
            const float* g = gwk + (((size_t)b * O + o) * T) * ldg + i;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (size_t)t * ldg);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p1[q][e] = fmaf(wv[e][t], gv[e], p1[q][e]);
                    gl[e][t] = fmaf(coef[e], gv[e], gl[e][t]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                partP[q] = fmaf(sv[e], p1[q][e], partP[q]);
                partR[q] = fmaf(vv[e], p1[q][e], partR[q]);
            }
        }
        if (DEMOD) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const float a = wave_sum(partP[q]), r = wave_sum(partR[q]);
                if (lane == 0) { red[1][wid][q] = a; red[2][wid][q] = r; }
            }
            __syncthreads();
            if (tid < 2 * NB) {
                tot[1][tid / NB][tid % NB] = red[1][2 * (tid / NB)][tid % NB] + red[1][2 * (tid / NB) + 1][tid % NB];
                tot[2][tid / NB][tid % NB] = red[2][2 * (tid / NB)][tid % NB] + red[2][2 * (tid / NB) + 1][tid % NB];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int b = 2 * q + half;
                if (b >= B || !live) continue;
                const float dv = d[(size_t)b * O + o];
                const float d3 = c3 * dv * dv * dv, d5 = c5 * dv * dv * dv * dv * dv;
                const float dA = tot[0][half][q], P = tot[1][half][q], R1 = tot[2][half][q];
                const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (size_t)b * I + i);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (size_t)b * I + i);
                const float ks = -d3 * R1 + 1.5f * d5 * dA * P;          // multiplies s (and s^2 in the W term)
                const float kv = -d3 * P;                                // multiplies v
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gs_acc[q][e] += w2[e] * (ks * sv[e] + kv * vv[e]) - 0.5f * d3 * dA * p1[q][e];
                    c2[e] += ks * sv[e] * sv[e] + 2.f * kv * sv[e] * vv[e];
                }
            }
        }
        // dL2/dW = gl + W * c2, both halves added
        if (half == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < T; ++t) xch[tid & 127][e * T + t] = fmaf(wv[e][t], c2[e], gl[e][t]);
        }
        __syncthreads();
        if (half == 0 && live) {
            float outv[4 * T];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < T; ++t) outv[e * T + t] = fmaf(wv[e][t], c2[e], gl[e][t]) + xch[tid][e * T + t];
#pragma unroll
            for (int j = 0; j < T; ++j)
                *reinterpret_cast<f32x4*>(gW + ((size_t)o * I + i) * T + 4 * j) =
                    f32x4{outv[4 * j], outv[4 * j + 1], outv[4 * j + 2], outv[4 * j + 3]};
        }
        __syncthreads();
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int b = 2 * q + half;
            if (b >= B) continue;
            *reinterpret_cast<f32x4*>(gs_part + ((size_t)og * B + b) * I + i) =
                f32x4{gs_acc[q][0], gs_acc[q][1], gs_acc[q][2], gs_acc[q][3]};
        }
    }
}

// gwk [B][O][taps][ldg] (the first backward's per-sample weight gradient), v [B][I] (cotangent of the style gradient) ->
// gW [O][I][taps], gs_part [ceil(O/o_group)][B][I] (both overwritten; the caller sums gs_part over its first axis).
// Limits: I % 4 == 0, I <= 512, taps in {1, 4, 9}, B <= 16, 16-byte aligned operands (else MSG_EUNSUPPORTED: the caller
// differentiates the composite formulation instead).
extern "C" int msg_modulate_backward2(const float* gwk, const float* W, const float* s, const float* d, const float* v,
                                      float* gW, float* gs_part, int B, int O, int I, int taps, int ldg, int o_group,
                                      float scale, void* stream) {
    if (B == 0) return MSG_OK;
    if (!gwk || !W || !s || !v || !gW || !gs_part || B < 0 || O <= 0 || I <= 0 || taps <= 0 || ldg < I || o_group <= 0)
        return MSG_EINVAL;
    if (I > 512 || I % 4 || ldg % 4 || B > MB_BMAX / 2 ||
        (((uintptr_t)gwk | (uintptr_t)W | (uintptr_t)s | (uintptr_t)v | (uintptr_t)gW | (uintptr_t)gs_part) & 15u))
        return MSG_EUNSUPPORTED;
    const int groups = (O + o_group - 1) / o_group;
    hipStream_t st = (hipStream_t)stream;
#define MB2(T_) do { if (d) hipLaunchKernelGGL((modulate_backward2_v4_kernel<true, T_>), dim3(groups), dim3(256), 0, st, gwk, W, s, d, v, \
                                               gW, gs_part, B, O, I, ldg, o_group, scale); \
                     else hipLaunchKernelGGL((modulate_backward2_v4_kernel<false, T_>), dim3(groups), dim3(256), 0, st, gwk, W, s, d, v, \
                                             gW, gs_part, B, O, I, ldg, o_group, scale); } while (0)
    switch (taps) {
        case 1: MB2(1); break;
        case 4: MB2(4); break;
        case 9: MB2(9); break;
        default: return MSG_EUNSUPPORTED;
    }
#undef MB2
    return MSG_CHECK_LAUNCH();
}

template <bool DEMOD>
static bool modulate_backward_v4_launch(const float* gwk, const float* W, const float* s, const float* d, float* gW,
                                        float* gs_part, int B, int O, int I, int taps, int ldg, int o_group, float scale,
                                        int groups, hipStream_t st) {
#define MB_V4(T_) hipLaunchKernelGGL((modulate_backward_v4_kernel<DEMOD, T_>), dim3(groups), dim3(256), 0, st, gwk, W, s, d, \
                                     gW, gs_part, B, O, I, ldg, o_group, scale); return true;
    switch (taps) {
        case 1: MB_V4(1)
        case 4: MB_V4(4)
        case 9: MB_V4(9)
        default: return false;
    }
#undef MB_V4
}

extern "C" int msg_modulate_backward(const float* gwk, const float* W, const float* s, const float* d, float* gW,
                                     float* gs_part, int B, int O, int I, int taps, int ldg, int o_group,
                                     float scale, void* stream) {
    if (B == 0) return MSG_OK;
    if (!gwk || !W || !s || !gW || !gs_part || B < 0 || O <= 0 || I <= 0 || taps <= 0 || ldg < I || o_group <= 0)
        return MSG_EINVAL;
    if (I > 256 * MB_SLOTS || taps > MB_TAPS || B > MB_BMAX / 2) return MSG_EUNSUPPORTED;
    const int groups = (O + o_group - 1) / o_group;
    hipStream_t st = (hipStream_t)stream;
    static const int v4 = msg_tunable("MSG_MODBWD_V4", 1);
    if (v4 && I % 4 == 0 && ldg % 4 == 0 && !(((uintptr_t)gwk | (uintptr_t)W | (uintptr_t)s | (uintptr_t)gW | (uintptr_t)gs_part) & 15u)) {
        const bool ok = d ? modulate_backward_v4_launch<true>(gwk, W, s, d, gW, gs_part, B, O, I, taps, ldg, o_group, scale, groups, st)
                          : modulate_backward_v4_launch<false>(gwk, W, s, d, gW, gs_part, B, O, I, taps, ldg, o_group, scale, groups, st);
        if (ok) return MSG_CHECK_LAUNCH();
    }
    if (d)
        hipLaunchKernelGGL((modulate_backward_kernel<true>), dim3(groups), dim3(256), 0, st, gwk, W, s, d, gW, gs_part,
                           B, O, I, taps, ldg, o_group, scale);
    else
        hipLaunchKernelGGL((modulate_backward_kernel<false>), dim3(groups), dim3(256), 0, st, gwk, W, s, d, gW, gs_part,
                           B, O, I, taps, ldg, o_group, scale);
    return MSG_CHECK_LAUNCH();
}
