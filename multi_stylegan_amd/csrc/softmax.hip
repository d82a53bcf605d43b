// a6: row softmax of the non-local block's attention map, and its backward.
//
//   y[r][:]  = softmax(x[r][:])                        (u_net_2d_discriminator.py:378: F.softmax(bmm(theta^T, phi), -1))
//   gx[r][:] = y[r][:] * (gy[r][:] - sum_c gy[r][c] * y[r][c])
//
// The map is [B * HW][HW / 4] (4096 x 1024 per sample at 256^2, 16384 x 4096 at 512^2) in the storage type; all
// arithmetic is fp32.  Pure streaming: ONE wave owns a row and keeps it in registers (16-byte chunk c of the row
// belongs to lane c % 64), so every element is read once and written once -- forward 2 x numel, backward 3 x numel
// bytes -- with two wave-shuffle reductions per row and no LDS.  Four rows per workgroup.
#include "msg_common.h"

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

template <typename T, int CH>      // CH = 16-byte chunks per lane
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows,
                                                           int cols) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / VEC;
    const uint4* xr = reinterpret_cast<const uint4*>(x + row * cols);
    uint4* yr = reinterpret_cast<uint4*>(y + row * cols);
    float v[CH][VEC];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        Vec16<T> t;
        if (c < nchunk) t.raw = xr[c];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[j][e] = c < nchunk ? t.get(e) : -INFINITY;
            m = fmaxf(m, v[j][e]);
        }
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[j][e] = __expf(v[j][e] - m);           // (-inf - m) -> 0 for the padding lanes
            s += v[j][e];
        }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        if (c >= nchunk) continue;
        Vec16<T> o;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, v[j][e] * inv);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, v[j][2 * e] * inv, v[j][2 * e + 1] * inv);
        }
        yr[c] = o.raw;
    }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ y, const T* __restrict__ gy,
                                                               T* __restrict__ gx, long long rows, int cols) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / VEC;
    const uint4* yr = reinterpret_cast<const uint4*>(y + row * cols);
    const uint4* gr = reinterpret_cast<const uint4*>(gy + row * cols);
    uint4* xr = reinterpret_cast<uint4*>(gx + row * cols);
    float yv[CH][VEC], gv[CH][VEC];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        Vec16<T> a, b;
        a.zero(); b.zero();
        if (c < nchunk) { a.raw = yr[c]; b.raw = gr[c]; }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            yv[j][e] = a.get(e);
            gv[j][e] = b.get(e);
            dot = fmaf(yv[j][e], gv[j][e], dot);
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        if (c >= nchunk) continue;
        Vec16<T> o;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, yv[j][e] * (gv[j][e] - dot));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o.set2(e, yv[j][2 * e] * (gv[j][2 * e] - dot), yv[j][2 * e + 1] * (gv[j][2 * e + 1] - dot));
        }
        xr[c] = o.raw;
    }
}

// Second-order terms of gx = y * (gy - <gy, y>) for a cotangent v of gx (R1 differentiates the discriminator's backward):
//   d_gy = y * (v - t),   d_y = v * (gy - s) - gy * t,     s = <gy, y>, t = <v, y>  per row.
// One pass: three maps read, two written (the torch formulation took ~15 elementwise / reduction passes over the
// [B, HW, HW/4] map: 2 GB per regularised iteration at 256^2).  The row stays in registers in its STORAGE type (raw 16-byte
// chunks, converted in both sweeps): 12 registers per chunk instead of 24 floats, so a 4096-column row fits without scratch.
template <typename T, int CH>
__global__ __launch_bounds__(256) void softmax_rows_bwd2_kernel(const T* __restrict__ y, const T* __restrict__ gy,
                                                                const T* __restrict__ v, T* __restrict__ d_y,
                                                                T* __restrict__ d_gy, long long rows, int cols) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / VEC;
    const uint4* yr = reinterpret_cast<const uint4*>(y + row * cols);
    const uint4* gr = reinterpret_cast<const uint4*>(gy + row * cols);
    const uint4* vr = reinterpret_cast<const uint4*>(v + row * cols);
    uint4* oy = reinterpret_cast<uint4*>(d_y + row * cols);
    uint4* og = reinterpret_cast<uint4*>(d_gy + row * cols);
    Vec16<T> ya[CH], ga[CH], va[CH];
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        ya[j].zero(); ga[j].zero(); va[j].zero();
        if (c < nchunk) { ya[j].raw = yr[c]; ga[j].raw = gr[c]; va[j].raw = vr[c]; }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float yy = ya[j].get(e);
            s = fmaf(ga[j].get(e), yy, s);
            t = fmaf(va[j].get(e), yy, t);
        }
    }
    s = wave_sum(s);
    t = wave_sum(t);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        if (c >= nchunk) continue;
        Vec16<T> o1, o2;
        float r1[VEC], r2[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float yy = ya[j].get(e), gg = ga[j].get(e), vv = va[j].get(e);
            r1[e] = vv * (gg - s) - gg * t;
            r2[e] = yy * (vv - t);
        }
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o1.set(e, r1[e]); o2.set(e, r2[e]); }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o1.set2(e, r1[2 * e], r1[2 * e + 1]); o2.set2(e, r2[2 * e], r2[2 * e + 1]); }
        }
        oy[c] = o1.raw;
        og[c] = o2.raw;
    }
}

static int softmax_chunks(int dtype, long long rows, int cols, const void* a, const void* b, const void* c) {
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    if (!a || !b || !c || rows < 0 || cols <= 0) return MSG_EINVAL;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (cols % vec || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15u)) return MSG_EUNSUPPORTED;
    const int per_lane = (cols / vec + 63) / 64;
    int ch = 1;
    while (ch < per_lane) ch *= 2;
    if (ch * vec > 64) return MSG_EUNSUPPORTED;          // a row must fit the wave's registers (4096 bf16 / f32 columns)
    return ch;
}

#define SOFTMAX_LAUNCH(KERNEL, T, ...)                                                                             \
    switch (ch) {                                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<T, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
        case 2: hipLaunchKernelGGL((KERNEL<T, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
        case 4: hipLaunchKernelGGL((KERNEL<T, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
        case 8: hipLaunchKernelGGL((KERNEL<T, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
        default: if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((KERNEL<T, 16>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
    }

extern "C" int msg_softmax_rows(const void* x, void* y, int dtype, long long rows, int cols, void* stream) {
    if (rows == 0) return MSG_OK;
    const int ch = softmax_chunks(dtype, rows, cols, x, y, y);
    if (ch < 0) return ch;
    const long long blocks = (rows + 3) / 4;
    if (blocks >= (1ll << 31)) return MSG_EUNSUPPORTED;
    dim3 grid((unsigned)blocks);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16) { SOFTMAX_LAUNCH(softmax_rows_kernel, bf16_t, (const bf16_t*)x, (bf16_t*)y, rows, cols) }
    else { SOFTMAX_LAUNCH(softmax_rows_kernel, float, (const float*)x, (float*)y, rows, cols) }
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_softmax_rows_backward(const void* y, const void* gy, void* gx, int dtype, long long rows, int cols,
                                         void* stream) {
    if (rows == 0) return MSG_OK;
    const int ch = softmax_chunks(dtype, rows, cols, y, gy, gx);
    if (ch < 0) return ch;
    const long long blocks = (rows + 3) / 4;
    if (blocks >= (1ll << 31)) return MSG_EUNSUPPORTED;
    dim3 grid((unsigned)blocks);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16) { SOFTMAX_LAUNCH(softmax_rows_bwd_kernel, bf16_t, (const bf16_t*)y, (const bf16_t*)gy, (bf16_t*)gx, rows, cols) }
    else { SOFTMAX_LAUNCH(softmax_rows_bwd_kernel, float, (const float*)y, (const float*)gy, (float*)gx, rows, cols) }
    return MSG_CHECK_LAUNCH();
}

/* ABI 5: second-order softmax terms in one pass (see softmax_rows_bwd2_kernel). */
extern "C" int msg_softmax_rows_backward2(const void* y, const void* gy, const void* v, void* d_y, void* d_gy, int dtype,
                                          long long rows, int cols, void* stream) {
    if (rows == 0) return MSG_OK;
    int ch = softmax_chunks(dtype, rows, cols, y, gy, v);
    if (ch < 0) return ch;
    const int ch2 = softmax_chunks(dtype, rows, cols, d_y, d_gy, d_gy);
    if (ch2 < 0) return ch2;
    const long long blocks = (rows + 3) / 4;
    if (blocks >= (1ll << 31)) return MSG_EUNSUPPORTED;
    dim3 grid((unsigned)blocks);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16) { SOFTMAX_LAUNCH(softmax_rows_bwd2_kernel, bf16_t, (const bf16_t*)y, (const bf16_t*)gy, (const bf16_t*)v, (bf16_t*)d_y, (bf16_t*)d_gy, rows, cols) }
    else { SOFTMAX_LAUNCH(softmax_rows_bwd2_kernel, float, (const float*)y, (const float*)gy, (const float*)v, (float*)d_y, (float*)d_gy, rows, cols) }
    return MSG_CHECK_LAUNCH();
}
