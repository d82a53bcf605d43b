// a5: the discriminator's pixel-wise head -- FusedLeakyReLU(C) followed by a bias-free 1x1 equalized conv to ONE plane
// (u_net_2d_discriminator.py:93-97 of the reference: `final_mapping`) -- as one streaming pass per direction.
//
// As two ops the head cost, per discriminator pass at 256^2 x 128 channels: the activation (read + write the map), the
// thin conv (read the activated map); backward: the conv's data gradient (write a map that is w[c] * g[pixel]), the
// activation backward (read that and the stored output, write), the conv's weight gradient (read the activated map) --
// six passes over a map whose information is ONE plane.  Here:
//     forward   y[p]     = wscale * sum_c w[c] * a[p][c],   a = lrelu(x[p][c] + b[c]) * scale          (reads x once)
//     backward  gx[p][c] = g[p] * wscale * w[c] * slope(x[p][c] + b[c]),  grad_b[c] = sum_p gx,  grad_w[c] = wscale * sum_p g[p] a[p][c]
//                                                                                                    (reads x once, writes gx)
// The activated map is never stored (fp32 inside the pass: one rounding less than the two-op form, which stored it in bf16).
// Layout as bias_act.hip's channels-last kernels: LC = C / 8 lanes across the channels of a pixel, 256 / LC pixels per
// step; the sums go through per-workgroup partial rows and the fixed-order reduce of bias_act.hip (deterministic).
#include "msg_common.h"

extern "C" int msg_bias_act_reduce_launch(const float* part_b, float* grad_bias, int C, long long n_b, const float* part_n,
                                          float* grad_nw, long long n_n, void* stream);

struct HeadParams {
    long long npix, ppb;           // pixels, pixels per workgroup
    int C, lanes_c;
    float alpha, scale, wscale;
};

__global__ __launch_bounds__(256) void pointwise_head_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ bias,
                                                                 const float* __restrict__ w, float* __restrict__ y,
                                                                 HeadParams p) {
    using V = Vec16<bf16_t>;
    const int lc = threadIdx.x % p.lanes_c, pl = threadIdx.x / p.lanes_c, npl = 256 / p.lanes_c;
    float bv[8], wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bv[e] = bias ? bias[lc * 8 + e] : 0.f; wv[e] = w[lc * 8 + e] * p.wscale; }
    const long long p0 = (long long)blockIdx.x * p.ppb;
    const long long p1 = (p0 + p.ppb < p.npix) ? p0 + p.ppb : p.npix;
    const float pos = p.scale, neg = p.scale * p.alpha;
    // two pixels per trip, both requested before either is used (as bias_act_bwd_cl_kernel)
    for (long long q = p0 + pl; q < p1; q += 2 * npl) {
        const long long qq[2] = {q, q + npl};
        const bool live1 = qq[1] < p1;
        V v[2];
        v[0].raw = *reinterpret_cast<const uint4*>(x + qq[0] * p.C + lc * 8);
        if (live1) v[1].raw = *reinterpret_cast<const uint4*>(x + qq[1] * p.C + lc * 8);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !live1) break;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[u].get(e) + bv[e];
                s = fmaf(wv[e], t * (t > 0.f ? pos : neg), s);
            }
            // the lanes of a pixel are consecutive: xor butterflies below lanes_c stay inside the pixel
            for (int off = p.lanes_c >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lc == 0) y[qq[u]] = s;
        }
    }
}

__global__ __launch_bounds__(256) void pointwise_head_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ bias,
                                                                 const float* __restrict__ w, const float* __restrict__ gy,
                                                                 bf16_t* __restrict__ gx, float* __restrict__ part,
                                                                 HeadParams p) {
    using V = Vec16<bf16_t>;
    __shared__ float red[256 * 16];
    const int lc = threadIdx.x % p.lanes_c, pl = threadIdx.x / p.lanes_c, npl = 256 / p.lanes_c;
    float bv[8], wv[8], sb[8], sw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = bias ? bias[lc * 8 + e] : 0.f;
        wv[e] = w[lc * 8 + e] * p.wscale;
        sb[e] = sw[e] = 0.f;
    }
    const long long p0 = (long long)blockIdx.x * p.ppb;
    const long long p1 = (p0 + p.ppb < p.npix) ? p0 + p.ppb : p.npix;
    const float pos = p.scale, neg = p.scale * p.alpha;
    for (long long q = p0 + pl; q < p1; q += 2 * npl) {
#pragma clang fp contract(off)
        const long long qq[2] = {q, q + npl};
        const bool live1 = qq[1] < p1;
        V v[2];
        float g[2] = {0.f, 0.f};
        v[0].raw = *reinterpret_cast<const uint4*>(x + qq[0] * p.C + lc * 8);
        g[0] = gy[qq[0]];
        if (live1) { v[1].raw = *reinterpret_cast<const uint4*>(x + qq[1] * p.C + lc * 8); g[1] = gy[qq[1]]; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !live1) break;
            V r;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[u].get(e) + bv[e];
                const float sl = t > 0.f ? pos : neg;
                f[e] = g[u] * wv[e] * sl;                   // d L / d (x + b)
                sb[e] += f[e];
                sw[e] += g[u] * (t * sl);                   // g * a  (the 1x1 conv's weight gradient, before wscale)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) r.set2(e, f[2 * e], f[2 * e + 1]);
            *reinterpret_cast<uint4*>(gx + qq[u] * p.C + lc * 8) = r.raw;
        }
    }
    // per-workgroup partial row [2 C]: bias sums, then weight sums; pixel lanes added in lane order
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[(pl * p.lanes_c + lc) * 16 + e] = sb[e];
        red[(pl * p.lanes_c + lc) * 16 + 8 + e] = sw[e];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < p.lanes_c * 16; j += 256) {
        const int lcj = j >> 4, k = j & 15;
        float s = 0.f;
        for (int r = 0; r < npl; ++r) s += red[(r * p.lanes_c + lcj) * 16 + k];
        part[(long long)blockIdx.x * 2 * p.C + (k >> 3) * p.C + lcj * 8 + (k & 7)] = s * (k >= 8 ? p.wscale : 1.f);
    }
}

static bool head_plan(long long npix, int C, HeadParams* p, long long* blocks) {
    if (npix <= 0 || C < 8 || C > 512 || C % 8) return false;
    int lanes_c = C / 8;
    if (lanes_c & (lanes_c - 1)) return false;             // the lanes of a pixel: a power of two (butterfly reduction)
    const long long npl = 256 / lanes_c;
    long long ppb = (npix + 2047) / 2048;                  // ~2048 workgroups
    ppb = ((ppb + npl - 1) / npl) * npl;
    p->npix = npix; p->ppb = ppb; p->C = C; p->lanes_c = lanes_c;
    *blocks = (npix + ppb - 1) / ppb;
    return *blocks < (1ll << 31);
}

extern "C" int msg_act_pointwise_head(const void* x, const float* bias, const float* w, float* y, int dtype,
                                      long long npix, int C, float wscale, float alpha, float scale, void* stream) {
    if (npix == 0) return MSG_OK;
    if (!x || !w || !y || npix < 0) return MSG_EINVAL;
    HeadParams p{};
    long long blocks;
    if (dtype != MSG_BF16 || !head_plan(npix, C, &p, &blocks) || (((uintptr_t)x) & 15u)) return MSG_EUNSUPPORTED;
    p.alpha = alpha; p.scale = scale; p.wscale = wscale;
    hipLaunchKernelGGL(pointwise_head_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       bias, w, y, p);
    return MSG_CHECK_LAUNCH();
}

extern "C" long long msg_act_pointwise_head_backward_workspace(long long npix, int C) {
    HeadParams p{};
    long long blocks;
    return head_plan(npix, C, &p, &blocks) ? blocks * 2 * C : 0;
}

extern "C" int msg_act_pointwise_head_backward(const void* x, const float* bias, const float* w, const float* gy, void* gx,
                                               float* grad_bias_and_w, int dtype, long long npix, int C, float wscale,
                                               float alpha, float scale, float* ws, long long ws_floats, void* stream) {
    if (npix == 0) {
        if (grad_bias_and_w && hipMemsetAsync(grad_bias_and_w, 0, sizeof(float) * 2 * C, (hipStream_t)stream) != hipSuccess)
            return MSG_ELAUNCH;
        return MSG_OK;
    }
    if (!x || !w || !gy || !gx || !grad_bias_and_w || npix < 0) return MSG_EINVAL;
    HeadParams p{};
    long long blocks;
    if (dtype != MSG_BF16 || !head_plan(npix, C, &p, &blocks) || ((((uintptr_t)x) | ((uintptr_t)gx)) & 15u)) return MSG_EUNSUPPORTED;
    if (!ws || ws_floats < blocks * 2 * C) return MSG_EINVAL;
    p.alpha = alpha; p.scale = scale; p.wscale = wscale;
    hipLaunchKernelGGL(pointwise_head_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       bias, w, gy, (bf16_t*)gx, ws, p);
    if (MSG_CHECK_LAUNCH() != MSG_OK) return MSG_ELAUNCH;
    // [blocks][2 C] partial rows -> grad_bias_and_w[0 .. C) = bias gradient, [C .. 2 C) = weight gradient
    return msg_bias_act_reduce_launch(ws, grad_bias_and_w, 2 * C, blocks, nullptr, nullptr, 0, stream);
}
