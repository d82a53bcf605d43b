// a3 / a1: the RGB skip path of the generator in one launch per level and direction ("1x1 ... with fused
// skip-upsample add"; reference multi_stylegan_generator.py:513-523 -- OutputBlock.forward: conv result + scalar bias +
// Upsample(skip), the latter an upfirdn2d with up = 2, the 4x4 FIR and pad (2, 1), :545-574).
//
// Before: four launches per level on tiny planar fp32 maps -- a cast / layout copy of the thin 1x1 conv's channels-last
// output, the bias add, the scalar generic FIR kernel on [B*C, h, w] planes (0.03-0.05 of HBM peak: 19 launches per
// iteration, the last generic FIR launches of the training step) and the sum.  Here: out[b,c,y,x] (fp32 planes, what the
// next level's skip and the image are) = float(conv[b,y,x,c]) + bias[c] + sum_taps fir * skip[b,c,.,.]; a thread owns
// four consecutive pixels of an output row for all C channels: one contiguous read of the conv result (4 pixels x ld
// channels), the 2 x 4 low-resolution neighbourhood per channel (the zero-insertion phase leaves 2 x 2 of the 4 x 4 taps
// per output pixel), one 16-byte store per channel plane.  HBM-bound: (ld * esz + 4 C + C) bytes per output pixel.
// Backward (the op is linear): one launch, a thread owns one LOW-resolution pixel: the gradient of the conv result for its
// 2 x 2 output pixels (cast, planes -> channels-last, padding channels zeroed) and the gradient of the skip, the
// transposed FIR as a GATHER over the 4 x 4 window of g around it (no scatter, no atomics: deterministic).  The
// second-order pass (path-length regulariser) is the forward applied to the cotangents.
#include "msg_common.h"

namespace {

struct RgbSkipParams {
    int B, C, H, W, ld;          // output map H x W; conv pixel pitch ld (elements); skip is [B, C, H/2, W/2]
    const float* fir;            // DEVICE pointer to the 4 x 4 taps (or NULL: no skip); wave-uniform loads
};

// FLIPPED taps: kf[ky][kx] = fir[3 - ky][3 - kx] (upfirdn2d is a true convolution, upfirdn2d_kernel.cu:114-133)
struct Taps {
    float kf[16];
    __device__ __forceinline__ explicit Taps(const float* fir) {
#pragma unroll
        for (int i = 0; i < 16; ++i) kf[i] = fir ? fir[15 - i] : 0.f;
    }
};

template <typename T> __device__ __forceinline__ float ld_elem(const T* p);
template <> __device__ __forceinline__ float ld_elem<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_elem<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st_elem(T* p, float v);
template <> __device__ __forceinline__ void st_elem<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_elem<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// Output rows / columns o and their two source positions: a = 0, 1 -> tap index (o & 1) + 2 a, source (o >> 1) - 1 + a + (o & 1)
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void rgb_skip_fwd_kernel(const T* __restrict__ conv, const float* __restrict__ bias,
                                                          const float* __restrict__ skip, float* __restrict__ out,
                                                          RgbSkipParams p) {
    const int w4 = p.W >> 2;
    const Taps tp(p.fir);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)p.B * p.H * w4) return;
    const int x0 = (int)(t % w4) * 4;
    const int oy = (int)((t / w4) % p.H);
    const int b = (int)(t / ((long long)w4 * p.H));
    const int h2 = p.H >> 1, w2 = p.W >> 1;
    const T* cp = conv + ((long long)(b * p.H + oy) * p.W + x0) * p.ld;
    const int py = oy & 1;
    const int sy0 = (oy >> 1) - 1 + py;                       // rows sy0, sy0 + 1 with taps ky = py, py + 2
    const int sx0 = (x0 >> 1) - 1;                            // columns sx0 .. sx0 + 3 serve the four pixels
    // the conv result of the four pixels: one 16-byte load per pixel when a pixel IS 16 bytes (bf16, pitch 8: the thin 1x1
    // kernel's output), element loads otherwise
    float cv[4][MAXC];
    if (sizeof(T) == 2 && p.ld == 8 && !((uintptr_t)conv & 15u)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vec16<bf16_t> q;
            q.raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(cp) + e * 8);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) cv[e][c] = q.get(c);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < MAXC; ++c) cv[e][c] = c < p.C ? ld_elem<T>(cp + e * p.ld + c) : 0.f;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c >= p.C) break;
        float v[4];
        const float bc = bias ? bias[c] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cv[e][c] + bc;
        if (skip) {
            const float* sp = skip + ((long long)b * p.C + c) * h2 * w2;
            float up[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int sy = sy0 + a, ky = py + 2 * a;
                if ((unsigned)sy >= (unsigned)h2) continue;
                float s[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = (unsigned)(sx0 + k) < (unsigned)w2 ? sp[(long long)sy * w2 + sx0 + k] : 0.f;
                // pixel e (parity e & 1): columns (e >> 1) + (e & 1) + {0, 1} of s, taps kx = (e & 1) + {0, 2}
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int px = e & 1, k0 = (e >> 1) + px;
                    up[e] = fmaf(s[k0], tp.kf[ky * 4 + px], up[e]);
                    up[e] = fmaf(s[k0 + 1], tp.kf[ky * 4 + px + 2], up[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += up[e];
        }
        *reinterpret_cast<f32x4*>(out + (((long long)b * p.C + c) * p.H + oy) * p.W + x0) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void rgb_skip_bwd_kernel(const float* __restrict__ g, T* __restrict__ g_conv,
                                                          float* __restrict__ g_skip, RgbSkipParams p) {
    const int h2 = p.H >> 1, w2 = p.W >> 1;
    const Taps tp(p.fir);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)p.B * h2 * w2) return;
    const int sx = (int)(t % w2), sy = (int)((t / w2) % h2), b = (int)(t / ((long long)w2 * h2));
    // the 4 x 4 window of g that met skip[sy, sx]: rows oy = 2 sy + 2 - ky, columns ox = 2 sx + 2 - kx, ky, kx = 0..3
    const int oy_hi = 2 * sy + 2, ox_hi = 2 * sx + 2;
    float ownv[4][MAXC];                                      // g at this thread's 2 x 2 output pixels, per channel
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ownv[q][c] = 0.f;
        if (c >= p.C) continue;
        const float* gp = g + ((long long)b * p.C + c) * p.H * p.W;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int oy = oy_hi - ky;
            if ((unsigned)oy >= (unsigned)p.H) continue;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int ox = ox_hi - kx;
                if ((unsigned)ox >= (unsigned)p.W) continue;
                const float gv = gp[(long long)oy * p.W + ox];
                acc = fmaf(gv, tp.kf[ky * 4 + kx], acc);
                if (ky >= 1 && ky <= 2 && kx >= 1 && kx <= 2) ownv[(2 - ky) * 2 + (2 - kx)][c] = gv;   // rows 2 sy, 2 sy + 1 / columns 2 sx, 2 sx + 1
            }
        }
        if (g_skip) g_skip[(((long long)b * p.C + c) * h2 + sy) * w2 + sx] = acc;
    }
    if (!g_conv) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        T* dst = g_conv + ((long long)(b * p.H + 2 * sy + (q >> 1)) * p.W + 2 * sx + (q & 1)) * p.ld;
        if (sizeof(T) == 2 && p.ld == 8 && !((uintptr_t)g_conv & 15u)) {                 // a pixel is one 16-byte store
            Vec16<bf16_t> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, ownv[q][2 * e], ownv[q][2 * e + 1]);
            *reinterpret_cast<uint4*>(dst) = o.raw;
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < p.ld) st_elem<T>(dst + c, ownv[q][c]);
            for (int c = MAXC; c < p.ld; ++c) st_elem<T>(dst + c, 0.f);                      // (padding channels: zeros, not stale bits)
        }
    }
}

bool fill(RgbSkipParams& p, const float* fir, int B, int C, int H, int W, int ld) {
    if (B <= 0 || C <= 0 || C > 8 || H <= 0 || W <= 0 || (W & 3) || (H & 1) || ld < C) return false;
    p = RgbSkipParams{B, C, H, W, ld, fir};
    return true;
}

}  // namespace

extern "C" int msg_rgb_skip_merge(const void* conv, int dtype, int ld, const float* bias, const float* skip,
                                  const float* fir, float* out, int B, int C, int H, int W, void* stream) {
    if (B == 0) return MSG_OK;
    if (!conv || !out || (skip && !fir)) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    RgbSkipParams p;
    if (!fill(p, fir, B, C, H, W, ld)) return MSG_EUNSUPPORTED;
    if (((uintptr_t)out & 15u) || ((uintptr_t)conv & 3u)) return MSG_EUNSUPPORTED;
    const long long threads = (long long)B * H * (W / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((rgb_skip_fwd_kernel<bf16_t, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)conv, bias, skip, out, p);
    else
        hipLaunchKernelGGL((rgb_skip_fwd_kernel<float, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)conv, bias, skip, out, p);
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_rgb_skip_merge_backward(const float* g, void* g_conv, int dtype, int ld, float* g_skip,
                                           const float* fir, int B, int C, int H, int W, void* stream) {
    if (B == 0) return MSG_OK;
    if (!g || (!g_conv && !g_skip) || (g_skip && !fir)) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    RgbSkipParams p;
    if (!fill(p, fir, B, C, H, W, ld)) return MSG_EUNSUPPORTED;
    const long long threads = (long long)B * (H / 2) * (W / 2);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((rgb_skip_bwd_kernel<bf16_t, 8>), grid, dim3(256), 0, (hipStream_t)stream, g, (bf16_t*)g_conv, g_skip, p);
    else
        hipLaunchKernelGGL((rgb_skip_bwd_kernel<float, 8>), grid, dim3(256), 0, (hipStream_t)stream, g, (float*)g_conv, g_skip, p);
    return MSG_CHECK_LAUNCH();
}
