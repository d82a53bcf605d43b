// Weight re-layout for the contraction kernels: ONE pass over a parameter produces every K-contiguous image the
// forward / data-gradient / modulation kernels read from it.
//
//   w     [O][I][T]  fp32, the parameter as the reference stores it (T = kh*kw taps, tap fastest)
//   fwd   [O][T][Ck] (or [T][O][Ck] when t_major: the 2x2 transposed conv's 4*O output rows)   = gain * w, zero for i >= I
//   dgrad [I][T][Ok] with the taps flipped (or not: transposed conv)                            = gain * w, zero for o >= O
//   wsq   [O][I]     fp32 sum over taps of w^2 (demodulation)
//
// Every output is optional.  Before this kernel each image was a torch transpose-copy (plus a zero fill, plus a scale),
// ~330 small launches and ~5 ms per training step, re-done after every optimiser step because the weights changed.
// A workgroup moves a 32 (o) x 32 (i) x T tile through LDS: the read is contiguous along (i, t), the forward image is
// written contiguous along i and the gradient image contiguous along o.
#include "msg_common.h"
#include <stdlib.h>

constexpr int RL_T = 16;                      // max taps held per tile (kernels up to 4x4)

// 16 bytes of storage elements from fp32 values (rounded like store_from_f32)
template <typename TO> __device__ __forceinline__ void store_vec16(TO* dst, const float* f);
template <> __device__ __forceinline__ void store_vec16<float>(float* dst, const float* f) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{f[0], f[1], f[2], f[3]};
}
template <> __device__ __forceinline__ void store_vec16<bf16_t>(bf16_t* dst, const float* f) {
    u32x4 pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = (uint32_t)f2bf(f[2 * k]) | ((uint32_t)f2bf(f[2 * k + 1]) << 16);
    *reinterpret_cast<u32x4*>(dst) = pk;
}

template <typename TO, int TC>       // TC: compile-time tap count (1, 4, 9, 16: index divisions become shifts / multiplies), 0 = runtime
__global__ __launch_bounds__(256) void relayout_weight_kernel(const float* __restrict__ w, TO* __restrict__ fwd,
                                                              TO* __restrict__ dgr, float* __restrict__ wsq,
                                                              int O, int I, int Trt, int Ck, int Ok, int flip, int t_major,
                                                              float gain) {
    const int T = TC ? TC : Trt;
    __shared__ float tile[32][32 * RL_T + 1];
    const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tid = threadIdx.x;
    const int row_len = 32 * T;                                   // floats of one o row of the tile (i-major, tap fastest)
    // ---- load (zero outside the parameter: those cells become the padding of the images)
    for (int e = tid; e < 32 * row_len; e += 256) {
        const int ro = e / row_len, c = e - ro * row_len;         // c = il * T + t
        const int o = o0 + ro, il = c / T;
        const int i = i0 + il;
        tile[ro][c] = (o < O && i < I) ? w[((size_t)o * I + i) * T + (c - il * T)] : 0.f;
    }
    __syncthreads();
    // ---- forward image: a thread writes ONE 16-byte vector of consecutive input channels (round 4: the 2-byte stores of the
    // first version -- 72 store instructions per thread and image -- made this a 36-us launch for 19 MB, 21 times per iteration)
    constexpr int VW = 16 / sizeof(TO);                           // elements per 16-byte store: 8 (bf16) / 4 (f32)
    constexpr int VPR = 32 / VW;                                  // vectors per 32-channel tile row
    const bool vec_ok = ((Ck % VW) | (Ok % VW)) == 0 && ((((uintptr_t)fwd | (uintptr_t)dgr) & 15u) == 0);
    if (fwd) {
        if (vec_ok) {
            for (int e = tid; e < 32 * T * VPR; e += 256) {
                const int v = e % VPR, rt = e / VPR;               // rt = ro * T + t
                const int ro = rt / T, t = rt - ro * T;
                const int o = o0 + ro, i = i0 + v * VW;
                if (o < O && i < Ck) {                             // (Ck % VW == 0: the vector is inside the row or outside it)
                    float f[VW];
#pragma unroll
                    for (int k = 0; k < VW; ++k) f[k] = tile[ro][(v * VW + k) * T + t] * gain;
                    const size_t row = t_major ? (size_t)t * O + o : (size_t)o * T + t;
                    store_vec16<TO>(fwd + row * Ck + i, f);
                }
            }
        } else {
            for (int e = tid; e < 32 * T * 32; e += 256) {
                const int il = e & 31, rt = e >> 5;                   // rt = ro * T + t
                const int ro = rt / T, t = rt - ro * T;
                const int o = o0 + ro, i = i0 + il;
                if (o < O && i < Ck) {
                    const size_t row = t_major ? (size_t)t * O + o : (size_t)o * T + t;
                    store_from_f32(fwd + row * Ck + i, tile[ro][il * T + t] * gain);
                }
            }
        }
    }
    // ---- data-gradient image: vectors of consecutive OUTPUT channels
    if (dgr) {
        if (vec_ok) {
            for (int e = tid; e < 32 * T * VPR; e += 256) {
                const int v = e % VPR, it = e / VPR;               // it = il * T + t
                const int il = it / T, t = it - il * T;
                const int o = o0 + v * VW, i = i0 + il;
                if (i < I && o < Ok) {
                    float f[VW];
#pragma unroll
                    for (int k = 0; k < VW; ++k) f[k] = tile[v * VW + k][il * T + t] * gain;
                    const int td = flip ? T - 1 - t : t;
                    store_vec16<TO>(dgr + ((size_t)i * T + td) * Ok + o, f);
                }
            }
        } else {
            for (int e = tid; e < 32 * T * 32; e += 256) {
                const int ro = e & 31, it = e >> 5;                   // it = il * T + t
                const int il = it / T, t = it - il * T;
                const int o = o0 + ro, i = i0 + il;
                if (i < I && o < Ok) {
                    const int td = flip ? T - 1 - t : t;
                    store_from_f32(dgr + ((size_t)i * T + td) * Ok + o, tile[ro][il * T + t] * gain);
                }
            }
        }
    }
    if (wsq) {
        for (int e = tid; e < 32 * 32; e += 256) {
            const int il = e & 31, ro = e >> 5;
            const int o = o0 + ro, i = i0 + il;
            if (o < O && i < I) {
                float s = 0.f;
                for (int t = 0; t < T; ++t) { const float v = tile[ro][il * T + t]; s = fmaf(v, v, s); }
                wsq[(size_t)o * I + i] = s;
            }
        }
    }
}

extern "C" int msg_relayout_weight(const float* w, void* fwd, void* dgrad, float* wsq, int dtype,
                                   int O, int I, int T, int Ck, int Ok, int flip, int t_major, float gain,
                                   void* stream) {
    if (!w || O <= 0 || I <= 0 || T <= 0 || (fwd && Ck < I) || (dgrad && Ok < O)) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    if (T > RL_T) return MSG_EUNSUPPORTED;
    // the tile grid must also cover the zero padding of the images (i up to Ck, o up to Ok)
    const int o_ext = dgrad && Ok > O ? Ok : O, i_ext = fwd && Ck > I ? Ck : I;
    dim3 grid((o_ext + 31) / 32, (i_ext + 31) / 32);
    if (grid.y > 65535) return MSG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define RL_LAUNCH(TC_)                                                                                                 \
    do {                                                                                                               \
        if (dtype == MSG_BF16)                                                                                         \
            hipLaunchKernelGGL((relayout_weight_kernel<bf16_t, TC_>), grid, dim3(256), 0, s, w, (bf16_t*)fwd,          \
                               (bf16_t*)dgrad, wsq, O, I, T, Ck, Ok, flip, t_major, gain);                             \
        else                                                                                                           \
            hipLaunchKernelGGL((relayout_weight_kernel<float, TC_>), grid, dim3(256), 0, s, w, (float*)fwd,            \
                               (float*)dgrad, wsq, O, I, T, Ck, Ok, flip, t_major, gain);                              \
    } while (0)
    switch (T) {
        case 1: RL_LAUNCH(1); break;
        case 4: RL_LAUNCH(4); break;
        case 9: RL_LAUNCH(9); break;
        case 16: RL_LAUNCH(16); break;
        default: RL_LAUNCH(0); break;
    }
#undef RL_LAUNCH
    return MSG_CHECK_LAUNCH();
}

// ---- tap gathering for convolutions with very few input channels (the discriminator's first layer: 6 channels) -------
// y[b][h][w][t*C + c] = x[b][h + kh_t - pad][w + kw_t - pad][c]   (zero outside the image and for t*C + c >= taps*C)
// turns a kh x kw conv over C channels into a 1x1 conv over Ko = one 128-byte run of (tap, channel) pairs: the
// implicit-GEMM kernels then spend ONE K-step per output tile instead of kh*kw steps that are 90 % zero padding, and
// the weight gradient reads gy once instead of once per tap.
template <typename T>
__global__ __launch_bounds__(256) void gather_taps_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W,
                                                          int Cx, int C, int kh, int kw, int pad, int Ko) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const int vpp = Ko / VEC;                                     // output vectors per pixel (8: divides the grid stride)
    const long long total = (long long)B * H * W * vpp;
    const long long first = (long long)blockIdx.x * 256 + threadIdx.x;
    // the thread's vector slot inside a pixel never changes over its grid-stride loop: its (tap, channel) pairs are
    // decoded ONCE
    const int v = (int)(first % vpp);
    int dy[VEC], dx[VEC], cc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int k = v * VEC + e;
        const int t = k / C;
        cc[e] = (t < kh * kw) ? k - t * C : -1;
        dy[e] = t / kw - pad;
        dx[e] = t - (t / kw) * kw - pad;
    }
    // vpp divides the grid stride (host), so the thread walks PIXELS with a constant stride: one 64-bit division up
    // front, 32-bit ones in the loop
    const unsigned npix = (unsigned)(total / vpp), pstep = (unsigned)((long long)gridDim.x * 256 / vpp);
    for (unsigned pix = (unsigned)(first / vpp); pix < npix; pix += pstep) {
        const long long vi = (long long)pix * vpp + v;
        const unsigned row = pix / (unsigned)W;
        const int w_ = (int)(pix - row * (unsigned)W);
        const int b = (int)(row / (unsigned)H);
        const int h_ = (int)(row - (unsigned)b * (unsigned)H);
        const T* xb = x + (long long)b * H * W * Cx;
        T out[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int ih = h_ + dy[e], iw = w_ + dx[e];
            T val = 0;
            if (cc[e] >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                val = xb[((long long)ih * W + iw) * Cx + cc[e]];
            out[e] = val;
        }
        *reinterpret_cast<uint4*>(y + vi * VEC) = *reinterpret_cast<const uint4*>(out);
    }
}

// The same for the case that matters (the discriminator's first layer: 3x3, pad 1, <= 8 channels stored as ONE 16-byte
// vector per pixel, bf16, map width a multiple of 32): a workgroup stages the 3 x 34 input vectors that 32 consecutive
// pixels of an image row need in LDS -- 102 coalesced 16-byte loads -- and every thread assembles one output vector from
// eight 2-byte LDS reads.  The generic kernel above issues eight 2-byte GLOBAL loads per output vector and is bound by their
// issue rate (1.7 TB/s of stores on 6 channels @256^2); this one by its stores.
__global__ __launch_bounds__(256) void gather_taps3x3_lds_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H,
                                                                 int W, int C, long long n_seg) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[3 * 34 * 8];
    const int tid = threadIdx.x, p = tid >> 3, v = tid & 7;
    int off[8];                                                  // LDS element offset of this thread's 8 (tap, channel) pairs, -1: zero
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = v * 8 + e, t = k / C, c = k - t * C;
        off[e] = t < 9 ? ((t / 3) * 34 + p + t % 3) * 8 + c : -1;
    }
    const int segs_per_row = W / 32;
    for (long long seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const long long row = seg / segs_per_row;                // (b, h) row of the batch
        const int w0 = (int)(seg - row * segs_per_row) * 32;
        const int h = (int)(row % H);
        __syncthreads();                                         // (the previous segment's reads are done)
        if (tid < 102) {
            const int r = tid / 34, col = tid - r * 34;
            const int ih = h + r - 1, iw = w0 + col - 1;
            uint4 val = make_uint4(0, 0, 0, 0);
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                val = *reinterpret_cast<const uint4*>(x + ((row - h + ih) * W + iw) * 8);
            *reinterpret_cast<uint4*>(tile + tid * 8) = val;
        }
        __syncthreads();
        unsigned int o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = off[2 * e] >= 0 ? tile[off[2 * e]] : 0u, hi = off[2 * e + 1] >= 0 ? tile[off[2 * e + 1]] : 0u;
            o[e] = lo | (hi << 16);
        }
        *reinterpret_cast<uint4*>(y + ((row * W + w0 + p) * 8 + v) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int msg_gather_taps(const void* x, void* y, int dtype, int B, int H, int W, int Cx, int C, int kh, int kw,
                               int pad, int Ko, void* stream) {
    if (B == 0) return MSG_OK;
    if (!x || !y || B < 0 || H <= 0 || W <= 0 || C <= 0 || Cx < C || kh <= 0 || kw <= 0 || Ko < kh * kw * C) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (Ko % vec || 256 % (Ko / vec) || (((uintptr_t)y) & 15u)) return MSG_EUNSUPPORTED;
    if ((long long)B * H * W >= (1ll << 31)) return MSG_EUNSUPPORTED;
    const long long total = (long long)B * H * W * (Ko / vec);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    static const int lds_path = msg_tunable("MSG_GATHER_LDS", 1);
    if (lds_path && dtype == MSG_BF16 && kh == 3 && kw == 3 && pad == 1 && Cx == 8 && Ko == 64 && W % 32 == 0) {
        const long long n_seg = (long long)B * H * (W / 32);
        const unsigned nb = (unsigned)(n_seg < 16384 ? n_seg : 16384);
        hipLaunchKernelGGL(gather_taps3x3_lds_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, n_seg);
        return MSG_CHECK_LAUNCH();
    }
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((gather_taps_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W,
                           Cx, C, kh, kw, pad, Ko);
    else
        hipLaunchKernelGGL((gather_taps_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, Cx,
                           C, kh, kw, pad, Ko);
    return MSG_CHECK_LAUNCH();
}

// ABI 5.  The adjoint of msg_gather_taps (same-size kh x kw convs over a few channels, stride 1): g [B, H, W, Ko] with
// K index (tap * C + c) -> gx [B, H, W, ldx], gx[q, c] = sum_t g[q - (dy_t, dx_t), t * C + c] (+ add[q, c]), channels C..ldx-1
// zeroed.  With it the DATA gradient of the discriminator's first layer (3x3, 128 -> 6 channels @256^2) is a 1x1 contraction
// 128 -> 54 followed by this fold -- two streaming passes -- instead of nine K-steps per tile on a 128 x 128 MFMA tile that is 95 %
// padding (392 us at batch 16: 37 TFLOP/s).  One thread per output pixel, fp32 accumulation.
template <typename T>
__global__ __launch_bounds__(256) void fold_taps_kernel(const T* __restrict__ g, const T* __restrict__ add, T* __restrict__ gx,
                                                        int B, int H, int W, int Ko, int C, int kh, int kw, int pad, int ldx,
                                                        int ld_add) {
    const long long npix = (long long)B * H * W;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
        const long long row = pix / W;
        const int w_ = (int)(pix - row * W);
        const int h_ = (int)(row % H);
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = (add && c < C) ? load_as_f32(add + pix * ld_add + c) : 0.f;
        for (int t = 0; t < kh * kw; ++t) {
            const int dy = t / kw - pad, dx = t - (t / kw) * kw - pad;
            const int sh = h_ - dy, sw = w_ - dx;
            if ((unsigned)sh >= (unsigned)H || (unsigned)sw >= (unsigned)W) continue;
            const T* src = g + (pix - ((long long)dy * W + dx)) * Ko + t * C;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) acc[c] += load_as_f32(src + c);
        }
        T* dst = gx + pix * ldx;
        for (int c = 0; c < ldx; ++c) store_from_f32(dst + c, c < C && c < 8 ? acc[c] : 0.f);
    }
}

// The case that matters (3x3, pad 1, <= 8 channels, bf16, Ko = 64 i.e. ONE 128-byte run per pixel, map width a multiple of 32,
// 16-byte output pixels): the generic kernel above reads 12 bytes at nine scattered pixels per thread -- every wave load touches
// 64 cache lines -- and ran the 256^2 batch-16 map in 810 us.  Here a workgroup stages the 3 x 34 pixel runs that 32
// consecutive output pixels of an image row need (816 coalesced 16-byte loads), thread (pixel, channel) adds its nine taps
// from LDS, and 32 threads store one 16-byte pixel each.
__global__ __launch_bounds__(256) void fold_taps3x3_lds_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ add,
                                                               bf16_t* __restrict__ gx, int B, int H, int W, int C,
                                                               long long n_seg) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[3 * 34 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t outp[32 * 8];
    const int tid = threadIdx.x, p = tid >> 3, c = tid & 7;
    const int segs_per_row = W / 32;
    for (long long seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const long long row = seg / segs_per_row;                // (b, h) row of the batch
        const int w0 = (int)(seg - row * segs_per_row) * 32;
        const int h = (int)(row % H);
        __syncthreads();                                         // (the previous segment's reads are done)
        for (int v = tid; v < 3 * 34 * 8; v += 256) {
            const int px = v >> 3, r = px / 34, col = px - r * 34;
            const int ih = h + r - 1, iw = w0 + col - 1;
            uint4 val = make_uint4(0, 0, 0, 0);
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                val = *reinterpret_cast<const uint4*>(g + ((row - h + ih) * W + iw) * 64 + (v & 7) * 8);
            *reinterpret_cast<uint4*>(tile + v * 8) = val;
        }
        __syncthreads();
        float acc = 0.f;
        if (c < C) {
            if (add) acc = bf2f(add[(row * W + w0 + p) * 8 + c]);
            // gx[q, c] = sum_t g[q - (dy, dx), t * C + c]: source pixel (h - dy, w - dx) sits at tile row 1 - dy, column p + 1 - dx
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                acc += bf2f(tile[((1 - dy) * 34 + p + 1 - dx) * 64 + t * C + c]);
            }
        }
        outp[p * 8 + c] = f2bf(acc);
        __syncthreads();
        if (tid < 32) *reinterpret_cast<uint4*>(gx + (row * W + w0 + tid) * 8) = *reinterpret_cast<const uint4*>(outp + tid * 8);
    }
}

extern "C" int msg_fold_taps(const void* g, const void* add, void* gx, int dtype, int B, int H, int W, int Ko, int C, int kh,
                             int kw, int pad, int ldx, int ld_add, void* stream) {
    if (B == 0) return MSG_OK;
    if (!g || !gx || B < 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || Ko < kh * kw * C || ldx < C ||
        (add && ld_add < C))
        return MSG_EINVAL;
    if ((dtype != MSG_F32 && dtype != MSG_BF16) || C > 8 || (long long)B * H * W >= (1ll << 31)) return MSG_EUNSUPPORTED;
    const long long npix = (long long)B * H * W;
    const unsigned blocks = (unsigned)((npix + 255) / 256 < 65536 ? (npix + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16 && kh == 3 && kw == 3 && pad == 1 && Ko == 64 && ldx == 8 && W % 32 == 0 && (!add || ld_add == 8) &&
        !(((uintptr_t)g | (uintptr_t)gx | (uintptr_t)add) & 15u)) {
        const long long n_seg = (long long)B * H * (W / 32);
        const unsigned nb = (unsigned)(n_seg < 16384 ? n_seg : 16384);
        hipLaunchKernelGGL(fold_taps3x3_lds_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)add, (bf16_t*)gx,
                           B, H, W, C, n_seg);
        return MSG_CHECK_LAUNCH();
    }
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((fold_taps_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)add,
                           (bf16_t*)gx, B, H, W, Ko, C, kh, kw, pad, ldx, ld_add);
    else
        hipLaunchKernelGGL((fold_taps_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)g, (const float*)add,
                           (float*)gx, B, H, W, Ko, C, kh, kw, pad, ldx, ld_add);
    return MSG_CHECK_LAUNCH();
}
