// a6: the 2x2 / stride-2 max-pooling of the NonLocalBlock's key and value maps (u_net_2d_discriminator.py:366-370:
// F.max_pool2d on phi(x) and g(x)), channels-last.  The library kernels keep an int64 index per output element and their
// backward scatters through it at 0.5 TB/s; here the forward leaves TWO BITS per element (the winner of the window, first
// maximum in scan order as the library's `>` comparison picks it) packed into one 16-bit word per 16-byte channel vector,
// and the backward is one pass that writes each input vector once: gy where the element won, zero elsewhere.
#include "msg_common.h"

template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             unsigned short* __restrict__ idx, int B, int H, int W, int C,
                                                             long long ldx) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const int cv = C / VEC, OH = H / 2, OW = W / 2;
    const long long total = (long long)B * OH * OW * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long long t = i / cv;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long b = t / OH;
        const T* p = x + ((b * H + 2 * oh) * W + 2 * ow) * ldx + (long long)c * VEC;
        V v[4];
        v[0].raw = *reinterpret_cast<const uint4*>(p);
        v[1].raw = *reinterpret_cast<const uint4*>(p + ldx);
        v[2].raw = *reinterpret_cast<const uint4*>(p + (long long)W * ldx);
        v[3].raw = *reinterpret_cast<const uint4*>(p + (long long)W * ldx + ldx);
        float best[VEC];
        unsigned int sel = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float m = v[0].get(e);
            unsigned int w = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float f = v[k].get(e);
                if (f > m || f != f) { m = f; w = k; }          // (the library's rule: strictly greater, or NaN, takes over)
            }
            best[e] = m;
            sel |= w << (2 * e);
        }
        V o;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, best[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, best[2 * e], best[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(y + i * VEC) = o.raw;
        if (idx) idx[i] = (unsigned short)sel;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const T* __restrict__ gy, const unsigned short* __restrict__ idx,
                                                             T* __restrict__ gx, int B, int H, int W, int C) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const int cv = C / VEC, OH = H / 2, OW = W / 2;
    const long long total = (long long)B * OH * OW * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long long t = i / cv;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long b = t / OH;
        V g;
        g.raw = *reinterpret_cast<const uint4*>(gy + i * VEC);
        const unsigned int sel = idx[i];
        T* p = gx + ((b * H + 2 * oh) * W + 2 * ow) * (long long)C + (long long)c * VEC;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            V o;
            float f[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = ((sel >> (2 * e)) & 3u) == (unsigned)k ? g.get(e) : 0.f;
            if constexpr (VEC == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set(e, f[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
            }
            *reinterpret_cast<uint4*>(p + ((k >> 1) * (long long)W + (k & 1)) * C) = o.raw;
        }
    }
}

// The transpose of the backward's scatter: v [B, H, W, C] read at the window positions the forward chose -> [B, H/2, W/2, C].
// This is the derivative of the pooling's backward with respect to its incoming gradient (the map is linear in gy, the routing
// is fixed by x): what R1's second differentiation needs, instead of the library's pooling re-run on the saved input.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_gather_kernel(const T* __restrict__ v, const unsigned short* __restrict__ idx,
                                                                T* __restrict__ out, int B, int H, int W, int C, long long ldv) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const int cv = C / VEC, OH = H / 2, OW = W / 2;
    const long long total = (long long)B * OH * OW * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long long t = i / cv;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long b = t / OH;
        const T* p = v + ((b * H + 2 * oh) * W + 2 * ow) * ldv + (long long)c * VEC;
        V w[4];
        w[0].raw = *reinterpret_cast<const uint4*>(p);
        w[1].raw = *reinterpret_cast<const uint4*>(p + ldv);
        w[2].raw = *reinterpret_cast<const uint4*>(p + (long long)W * ldv);
        w[3].raw = *reinterpret_cast<const uint4*>(p + (long long)W * ldv + ldv);
        const unsigned int sel = idx[i];
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const unsigned k = (sel >> (2 * e)) & 3u;
            f[e] = k == 0 ? w[0].get(e) : (k == 1 ? w[1].get(e) : (k == 2 ? w[2].get(e) : w[3].get(e)));
        }
        V o;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, f[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(out + i * VEC) = o.raw;
    }
}

static int pool_check(const void* a, const void* b, int dtype, int B, int H, int W, int C) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0) return MSG_EINVAL;
    if (B == 0) return MSG_OK;
    if (!a || !b) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (H % 2 || W % 2 || C % vec || (((uintptr_t)a | (uintptr_t)b) & 15u)) return MSG_EUNSUPPORTED;
    return 1;                                                         // go on (status codes are <= 0)
}

// x [B, H, W, C] channels-last with pixel pitch ldx (>= C) -> y [B, H/2, W/2, C] dense; idx (may be NULL: no backward will follow)
// one 16-bit word per 16-byte vector of y.
extern "C" int msg_maxpool2x2_fwd(const void* x, void* y, unsigned short* idx, int dtype, int B, int H, int W, int C,
                                  long long ldx, void* stream) {
    const int rc = pool_check(x, y, dtype, B, H, W, C);
    if (rc <= 0) return rc;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (ldx < C || ldx % vec) return MSG_EINVAL;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / vec);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((maxpool2x2_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, B, H, W, C, ldx);
    else
        hipLaunchKernelGGL((maxpool2x2_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)y, idx, B, H, W, C, ldx);
    return MSG_CHECK_LAUNCH();
}

// gy [B, H/2, W/2, C] dense, idx from the forward -> gx [B, H, W, C] dense, every element written.
extern "C" int msg_maxpool2x2_bwd(const void* gy, const unsigned short* idx, void* gx, int dtype, int B, int H, int W, int C,
                                  void* stream) {
    const int rc = pool_check(gy, gx, dtype, B, H, W, C);
    if (rc <= 0) return rc;
    if (!idx) return MSG_EINVAL;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / vec);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((maxpool2x2_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)gy, idx, (bf16_t*)gx, B, H, W, C);
    else
        hipLaunchKernelGGL((maxpool2x2_bwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)gy, idx, (float*)gx, B, H, W, C);
    return MSG_CHECK_LAUNCH();
}

// ABI 5.  v [B, H, W, C] channels-last with pixel pitch ldv, idx from the forward -> out [B, H/2, W/2, C] dense: v at the winners.
extern "C" int msg_maxpool2x2_gather(const void* v, const unsigned short* idx, void* out, int dtype, int B, int H, int W, int C,
                                     long long ldv, void* stream) {
    const int rc = pool_check(v, out, dtype, B, H, W, C);
    if (rc <= 0) return rc;
    if (!idx) return MSG_EINVAL;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (ldv < C || ldv % vec) return MSG_EINVAL;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / vec);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((maxpool2x2_gather_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)v, idx, (bf16_t*)out, B, H, W, C, ldv);
    else
        hipLaunchKernelGGL((maxpool2x2_gather_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)v, idx, (float*)out, B, H, W, C, ldv);
    return MSG_CHECK_LAUNCH();
}
