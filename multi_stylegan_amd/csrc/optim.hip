// a7 (the adversarial step's tail): Adam over a FLAT fp32 parameter / gradient / moment store, optionally with the
// generator's exponential moving average in the same pass.
//
// The reference steps ~400 small parameter tensors per network with torch.optim.Adam (model_wrapper.py:296-300, :410-414)
// and then walks them again for the EMA (misc.py exponential_moving_average).  torch's fused multi-tensor Adam does that in
// a handful of launches on the device, but costs ~8 ms of HOST time per step (per-parameter step tensors, pointer
// tables, one call per parameter group) -- more than the device work it enqueues, so the GPU idles behind it.  The
// trainer keeps every parameter, its gradient and both moments as views into flat buffers (multi_stylegan_amd/optim.py),
// which turns a step into ONE launch per run of parameters that share their hyper-parameters:
//
//   g    = grad * coef[0]                  (coef: device scalar = gradient-clipping factor x 1 / world size)
//   m    = lerp(m, g, 1 - beta1)           (torch's exp_avg.lerp_(grad, 1 - beta1), in at::lerp's two-sided form)
//   v    = beta2 v + (1 - beta2) g g
//   p   -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)         bc_i = 1 - beta_i^step
//   ema  = decay ema + (1 - decay) p       (if ema != NULL)
//
// the arithmetic of torch/optim/adam.py's single-tensor path in the same order.  Pure streaming: 16 B per lane and
// array, 16 (20 with the EMA) bytes read and 12 (16) written per element.
#include "msg_common.h"

// at::lerp's two-sided form: exact at both ends (weight 1 -- the reference's beta1 = 0 -- returns b itself)
__device__ __forceinline__ float lerp_like_torch(float a, float b, float w) {
    return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

__global__ __launch_bounds__(256) void flat_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, long long n, const float* __restrict__ coef,
                                                        float step_size, float beta1, float beta2, float eps,
                                                        float inv_bc2_sqrt, float ema_decay) {
    const float c = coef ? coef[0] : 1.f;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float4 ev = ema ? reinterpret_cast<float4*>(ema)[i] : float4{0.f, 0.f, 0.f, 0.f};
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x; float* ep = &ev.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gp[e] * c;
            mp[e] = lerp_like_torch(mp[e], gr, 1.f - beta1);
            vp[e] = beta2 * vp[e] + (1.f - beta2) * gr * gr;
            pp[e] = pp[e] - step_size * (mp[e] / (sqrtf(vp[e]) * inv_bc2_sqrt + eps));
            ep[e] = ema_decay * ep[e] + (1.f - ema_decay) * pp[e];
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (ema) reinterpret_cast<float4*>(ema)[i] = ev;
    }
    // tail (n not a multiple of 4): the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        const float gr = g[i] * c;
        const float mm = lerp_like_torch(m[i], gr, 1.f - beta1);
        const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        const float pn = p[i] - step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
        m[i] = mm; v[i] = vv; p[i] = pn;
        if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * pn;
    }
}

__global__ __launch_bounds__(256) void flat_ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long long n,
                                                       float decay) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        ema[i] = decay * ema[i] + (1.f - decay) * p[i];
}

static unsigned flat_grid(long long n) {
    long long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    return (unsigned)(blocks > 2048 ? 2048 : blocks);
}

extern "C" int msg_flat_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                             long long n, const float* coef, float lr, float beta1, float beta2, float eps, int step,
                             float ema_decay, void* stream) {
    if (n == 0) return MSG_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return MSG_EINVAL;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema) & 15)
        return MSG_EUNSUPPORTED;                                 // flat stores are 16-byte aligned
    // bias corrections in double like the Python of torch.optim (its scalars are Python floats)
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(flat_adam_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, ema, n, coef, step_size, beta1, beta2, eps, inv_bc2_sqrt, ema_decay);
    return hipGetLastError() == hipSuccess ? MSG_OK : MSG_ELAUNCH;
}

extern "C" int msg_flat_ema(float* ema, const float* param, long long n, float decay, void* stream) {
    if (n == 0) return MSG_OK;
    if (!ema || !param || n < 0) return MSG_EINVAL;
    hipLaunchKernelGGL(flat_ema_kernel, dim3(flat_grid(n) * 4 > 2048 ? 2048 : flat_grid(n) * 4), dim3(256), 0,
                       (hipStream_t)stream, ema, param, n, decay);
    return hipGetLastError() == hipSuccess ? MSG_OK : MSG_ELAUNCH;
}
