// a3 on small maps: the pieces of the ACTIVATION-SCALING form of the modulated convolution,
//     conv(x, d_b * scale * W * s_b) = d_b * conv(s_b * x, scale * W)          (s per input channel, d per output channel)
// (multi_stylegan_generator.py:384-411 builds one weight set per sample; on the 4^2 .. 32^2 layers that form spends its time
// on the WEIGHTS -- 75 MB of per-sample weights written and read per 512-channel layer and pass, a 151 MB per-sample weight
// gradient and its fold -- for maps of 16 .. 1024 pixels).  Here the weights stay shared (batch folded into the contraction,
// cached kernel-side images) and the two scalings touch a few MB of activations:
//   msg_scale_reduce_channels:  out[b,p,c] = in[b,p,c] * v[b,c]   and, optionally,  red[b,c] = sum_p in[b,p,c] * other[b,p,c]
//       forward:  xs = x * s                      backward:  gc = g * d,  gd = sum_p g * c     and   gx = gxs * s,  gs = sum_p gxs * x
//   msg_scale_bias_act:         y = lrelu(c * d[b,n] + noise_w * noise[b,p] + bias[n]) * scale     (or just c * d[b,n])
// Maps are dense channels-last [B][P][C]; sums are per-workgroup, in a fixed order (deterministic).
#include "msg_common.h"

template <typename T>
__global__ __launch_bounds__(256) void scale_reduce_channels_kernel(const T* __restrict__ in, const T* __restrict__ other,
                                                                    const float* __restrict__ v, T* __restrict__ out,
                                                                    float* __restrict__ red, int P, int C) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    __shared__ float part[16][16][VEC];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int cv = blockIdx.x * 16 + cl, b = blockIdx.y, cvs = C / VEC;
    const bool live = cv < cvs;
    float sv[VEC], acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sv[e] = live ? v[(long long)b * C + cv * VEC + e] : 0.f; acc[e] = 0.f; }
    if (live) {
        const long long base = (long long)b * P * C + (long long)cv * VEC;
        for (int p = pl; p < P; p += 16) {
            V a, o, r;
            a.raw = *reinterpret_cast<const uint4*>(in + base + (long long)p * C);
            if (other) o.raw = *reinterpret_cast<const uint4*>(other + base + (long long)p * C);
            float f[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float x = a.get(e);
                f[e] = x * sv[e];
                if (other) acc[e] = fmaf(x, o.get(e), acc[e]);
            }
            if (out) {
                if constexpr (VEC == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r.set(e, f[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r.set2(e, f[2 * e], f[2 * e + 1]);
                }
                *reinterpret_cast<uint4*>(out + base + (long long)p * C) = r.raw;
            }
        }
    }
    if (!red) return;
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[pl][cl][e] = acc[e];
    __syncthreads();
    if (pl == 0 && live) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += part[k][cl][e];
            red[(long long)b * C + cv * VEC + e] = s;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_bias_act_kernel(const T* __restrict__ c, const float* __restrict__ d,
                                                             const float* __restrict__ bias, const float* __restrict__ noise,
                                                             const float* __restrict__ noise_w, T* __restrict__ y, int P, int C,
                                                             long long nvec, int noise_batch, int act_on, float alpha, float scale) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const int cvs = C / VEC;
    const float nw = noise ? noise_w[0] : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvs);
        const long long bp = i / cvs;
        const long long b = bp / P;
        V a, o;
        a.raw = *reinterpret_cast<const uint4*>(c + i * VEC);
        const float nz = noise ? nw * noise[noise_batch == 1 ? bp - b * P : bp] : 0.f;
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int n = cv * VEC + e;
            float val = a.get(e) * (d ? d[b * C + n] : 1.f);
            if (act_on) {
                val += nz + (bias ? bias[n] : 0.f);
                val = (val > 0.f ? val : val * alpha) * scale;
            }
            f[e] = val;
        }
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, f[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(y + i * VEC) = o.raw;
    }
}

static int small_check(int dtype, int B, int P, int C) {
    if (B < 0 || P <= 0 || C <= 0) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    if (C % (dtype == MSG_BF16 ? 8 : 4) || B > 65535) return MSG_EUNSUPPORTED;
    return MSG_OK;
}

// in / other / out: dense [B][P][C] in the storage type (other, out may be NULL); v [B][C] fp32; red [B][C] fp32 or NULL.
extern "C" int msg_scale_reduce_channels(const void* in, const void* other, const float* v, void* out, float* red, int dtype,
                                         int B, int P, int C, void* stream) {
    const int rc = small_check(dtype, B, P, C);
    if (rc != MSG_OK) return rc;
    if (B == 0) return MSG_OK;
    if (!in || !v || (!out && !red) || (red && !other) ||
        (((uintptr_t)in | (uintptr_t)other | (uintptr_t)out) & 15u)) return MSG_EINVAL;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    dim3 grid((C / vec + 15) / 16, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((scale_reduce_channels_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)other, v, (bf16_t*)out, red, P, C);
    else
        hipLaunchKernelGGL((scale_reduce_channels_kernel<float>), grid, dim3(256), 0, s, (const float*)in, (const float*)other, v, (float*)out, red, P, C);
    return MSG_CHECK_LAUNCH();
}

// y = act_on ? lrelu(c * d[b,n] + noise_w[0] * noise[b or 0, p] + bias[n], alpha) * scale : c * d[b,n];  c, y dense [B][P][C];
// d [B][C] fp32 or NULL (= 1); bias [C] / noise [noise_batch][P] / noise_w [1] fp32 or NULL.
extern "C" int msg_scale_bias_act(const void* c, const float* d, const float* bias, const float* noise, const float* noise_w,
                                  void* y, int dtype, int B, int P, int C, int noise_batch, int act_on, float alpha, float scale,
                                  void* stream) {
    const int rc = small_check(dtype, B, P, C);
    if (rc != MSG_OK) return rc;
    if (B == 0) return MSG_OK;
    if (!c || !y || (noise && (!noise_w || (noise_batch != 1 && noise_batch != B))) || (((uintptr_t)c | (uintptr_t)y) & 15u))
        return MSG_EINVAL;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    const long long nvec = (long long)B * P * (C / vec);
    const unsigned blocks = (unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((scale_bias_act_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)c, d, bias, noise, noise_w, (bf16_t*)y, P, C, nvec, noise_batch, act_on, alpha, scale);
    else
        hipLaunchKernelGGL((scale_bias_act_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)c, d, bias, noise, noise_w, (float*)y, P, C, nvec, noise_batch, act_on, alpha, scale);
    return MSG_CHECK_LAUNCH();
}
