// Minibatch standard deviation of the discriminator (reference multi_stylegan/u_net_2d_discriminator.py:205-217): the
// standard deviation over the batch at every (channel, pixel), averaged to ONE scalar, appended as an extra plane.
//   stat = mean_{c,h,w} sqrt(max(var_b(x[b,c,h,w]), alpha)),   y = cat([x, stat.expand(B,1,H,W)], dim=1)
// The reference is six elementwise / reduction passes plus the concatenation copy; here one kernel reads x once from
// HBM (each thread owns one 16-byte channel vector of one pixel and walks the samples of its group twice -- mean, then
// centred squares; the second walk is served by L1 / L2), writes the copy of x into the padded channels-last output
// (the concatenation) and leaves one partial sum per workgroup; a single-workgroup kernel adds the partials in a fixed
// order (deterministic) and writes the statistic plane.  `groups` independent batches may be concatenated along the batch
// axis (the trainer runs the real and the fake batch as one): each group gets its own statistic.
// Backward: gx = gy[:, :C] + gstat[g] * (x - mean) / (n std C H W) where var > alpha.  HBM-bound.
#include "msg_common.h"

namespace {

struct MbstdParams {
    int B, C, H, W, groups;
    int ldx, ldy;          // channel pitch (elements) of x / y
    float alpha;
};

// block partial sums of sqrt(max(var, alpha)); y[:, :C] = x
template <typename T>
__global__ __launch_bounds__(256) void mbstd_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        float* __restrict__ partial, MbstdParams p) {
    constexpr int VEC = 16 / sizeof(T);
    const int cv = p.C / VEC, n = p.B / p.groups;
    const long long items = (long long)p.H * p.W * cv;                  // per group
    const int g = blockIdx.y;
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    float local = 0.f;
    if (it < items) {
        const long long pix = it / cv;
        const int c0 = (int)(it - pix * cv) * VEC;
        const long long hw = (long long)p.H * p.W;
        const T* xs = x + ((long long)g * n * hw + pix) * p.ldx + c0;
        T* ys = y + ((long long)g * n * hw + pix) * p.ldy + c0;
        float mean[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) mean[e] = 0.f;
        for (int b = 0; b < n; ++b) {
            Vec16<T> v;
            v.raw = *reinterpret_cast<const uint4*>(xs + (long long)b * hw * p.ldx);
            *reinterpret_cast<uint4*>(ys + (long long)b * hw * p.ldy) = v.raw;
#pragma unroll
            for (int e = 0; e < VEC; ++e) mean[e] += v.get(e);
        }
        const float inv_n = 1.f / n;
#pragma unroll
        for (int e = 0; e < VEC; ++e) mean[e] *= inv_n;
        float var[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) var[e] = 0.f;
        for (int b = 0; b < n; ++b) {
            Vec16<T> v;
            v.raw = *reinterpret_cast<const uint4*>(xs + (long long)b * hw * p.ldx);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v.get(e) - mean[e]; var[e] = fmaf(d, d, var[e]); }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) local += sqrtf(fmaxf(var[e] * inv_n, p.alpha));
    }
    __shared__ float red[4];
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)g * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per group: stat = sum(partials) / (C H W) in a fixed order; plane y[b, :, :, C] = stat, pad channels = 0
template <typename T>
__global__ __launch_bounds__(256) void mbstd_plane_kernel(const float* __restrict__ partial, int nblocks,
                                                          float* __restrict__ stat, T* __restrict__ y, MbstdParams p) {
    const int g = blockIdx.x, n = p.B / p.groups;
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[(long long)g * nblocks + i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float value = red[0] / ((float)p.C * p.H * p.W);
    if (threadIdx.x == 0) stat[g] = value;
    const long long pixels = (long long)n * p.H * p.W;
    T* base = y + (long long)g * pixels * p.ldy;
    for (long long i = threadIdx.x; i < pixels; i += 256) {
        store_from_f32<T>(base + i * p.ldy + p.C, value);
        for (int c = p.C + 1; c < p.ldy; ++c) store_from_f32<T>(base + i * p.ldy + c, 0.f);
    }
}

// gx = gy[:, :C] + gstat[g] * (x - mean) / (n * std * C H W)  (zero where the variance was clamped)
template <typename T>
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                        const float* __restrict__ gstat, T* __restrict__ gx,
                                                        MbstdParams p, int ldgy, int ldgx) {
    constexpr int VEC = 16 / sizeof(T);
    const int cv = p.C / VEC, n = p.B / p.groups;
    const long long items = (long long)p.H * p.W * cv;
    const int g = blockIdx.y;
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= items) return;
    const long long pix = it / cv;
    const int c0 = (int)(it - pix * cv) * VEC;
    const long long hw = (long long)p.H * p.W;
    const long long first = (long long)g * n * hw + pix;
    const T* xs = x + first * p.ldx + c0;
    float mean[VEC], var[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) mean[e] = var[e] = 0.f;
    for (int b = 0; b < n; ++b) {
        Vec16<T> v;
        v.raw = *reinterpret_cast<const uint4*>(xs + (long long)b * hw * p.ldx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) mean[e] += v.get(e);
    }
    const float inv_n = 1.f / n;
#pragma unroll
    for (int e = 0; e < VEC; ++e) mean[e] *= inv_n;
    for (int b = 0; b < n; ++b) {
        Vec16<T> v;
        v.raw = *reinterpret_cast<const uint4*>(xs + (long long)b * hw * p.ldx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = v.get(e) - mean[e]; var[e] = fmaf(d, d, var[e]); }
    }
    const float k = gstat[g] * inv_n / ((float)p.C * p.H * p.W);
    float coef[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float vr = var[e] * inv_n;
        coef[e] = vr > p.alpha ? k * rsqrtf(vr) : 0.f;
    }
    for (int b = 0; b < n; ++b) {
        Vec16<T> v, gv, o;
        v.raw = *reinterpret_cast<const uint4*>(xs + (long long)b * hw * p.ldx);
        gv.raw = *reinterpret_cast<const uint4*>(gy + (first + (long long)b * hw) * ldgy + c0);
        float r[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[e] = fmaf(coef[e], v.get(e) - mean[e], gv.get(e));
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, r[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, r[2 * e], r[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(gx + (first + (long long)b * hw) * ldgx + c0) = o.raw;
    }
}

template <typename T>
int run_fwd(const void* x, void* y, float* stat, float* workspace, const MbstdParams& p, hipStream_t s) {
    constexpr int VEC = 16 / sizeof(T);
    const long long items = (long long)p.H * p.W * (p.C / VEC);
    const int nblocks = (int)((items + 255) / 256);
    hipLaunchKernelGGL((mbstd_fwd_kernel<T>), dim3(nblocks, p.groups), dim3(256), 0, s, (const T*)x, (T*)y, workspace, p);
    if (hipGetLastError() != hipSuccess) return MSG_ELAUNCH;
    hipLaunchKernelGGL((mbstd_plane_kernel<T>), dim3(p.groups), dim3(256), 0, s, workspace, nblocks, stat, (T*)y, p);
    return MSG_CHECK_LAUNCH();
}

template <typename T>
int run_bwd(const void* x, const void* gy, const float* gstat, void* gx, const MbstdParams& p, int ldgy, int ldgx,
            hipStream_t s) {
    constexpr int VEC = 16 / sizeof(T);
    const long long items = (long long)p.H * p.W * (p.C / VEC);
    hipLaunchKernelGGL((mbstd_bwd_kernel<T>), dim3((unsigned)((items + 255) / 256), p.groups), dim3(256), 0, s,
                       (const T*)x, (const T*)gy, gstat, (T*)gx, p, ldgy, ldgx);
    return MSG_CHECK_LAUNCH();
}

bool params_ok(const MbstdParams& p, int vec) {
    return p.B > 0 && p.C > 0 && p.H > 0 && p.W > 0 && p.groups > 0 && p.B % p.groups == 0 && p.C % vec == 0 &&
           p.ldx % vec == 0 && p.ldy % vec == 0 && p.ldx >= p.C && p.ldy > p.C;
}

}  // namespace

extern "C" long long msg_minibatch_stddev_workspace(int C, int H, int W, int groups, int dtype) {
    const int vec = dtype == MSG_F32 ? 4 : 8;
    const long long items = (long long)H * W * (C / vec);
    return ((items + 255) / 256) * groups;          // floats
}

extern "C" int msg_minibatch_stddev(const void* x, void* y, float* stat, float* workspace, int dtype,
                                    int B, int C, int H, int W, int ldx, int ldy, int groups, float alpha, void* stream) {
    if (!x || !y || !stat || !workspace) return MSG_EINVAL;
    MbstdParams p{B, C, H, W, groups, ldx, ldy, alpha};
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    if (!params_ok(p, dtype == MSG_F32 ? 4 : 8)) return MSG_EUNSUPPORTED;
    return dtype == MSG_F32 ? run_fwd<float>(x, y, stat, workspace, p, (hipStream_t)stream)
                            : run_fwd<bf16_t>(x, y, stat, workspace, p, (hipStream_t)stream);
}

extern "C" int msg_minibatch_stddev_backward(const void* x, const void* gy, const float* gstat, void* gx, int dtype,
                                             int B, int C, int H, int W, int ldx, int ldgy, int ldgx, int groups,
                                             float alpha, void* stream) {
    if (!x || !gy || !gstat || !gx) return MSG_EINVAL;
    MbstdParams p{B, C, H, W, groups, ldx, ldgy, alpha};
    const int vec = dtype == MSG_F32 ? 4 : 8;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    if (!params_ok(p, vec) || ldgx % vec || ldgx < C) return MSG_EUNSUPPORTED;
    return dtype == MSG_F32 ? run_bwd<float>(x, gy, gstat, gx, p, ldgy, ldgx, (hipStream_t)stream)
                            : run_bwd<bf16_t>(x, gy, gstat, gx, p, ldgy, ldgx, (hipStream_t)stream);
}
