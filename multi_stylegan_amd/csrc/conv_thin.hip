// a3/a4: the 1x1 convolutions with (almost) no channels on one side -- HBM-streaming kernels (bf16).
//
// The models hold a handful of pointwise convolutions whose contraction is negligible and whose time is one pass over a
// full-resolution map: the RGB heads of the generator (512 -> 3 + 3 per sample, multi_stylegan_generator.py:472-526) and
// their data gradient (6 -> 512), the discriminator's first residual conv (6 -> 128, u_net_2d_discriminator.py:170-175),
// its pixel-wise head (128 -> 1, :93-97) and that head's data gradient (1 -> 128).  On the 128 x 128 MFMA tile of
// conv_fprop.hip they ran at 2.4 - 3.4 TB/s: a tile that is 95 % padding still stages, barriers and drains like a full
// one.  Here nothing is staged:
//
//   thin N (N <= 8 outputs):  a wave owns 16 pixels; every lane loads 16 B of a pixel's channels straight into the
//     B operand of v_mfma_f32_16x16x32_bf16 (pixel = lane & 15, 8 channels at slot lane >> 4), the weights -- 16 rows,
//     the real ones first, zeros behind -- stay in registers as the A operand; the accumulator D[n][pixel] gives lanes
//     0..31 the 4 + 4 output channels of their pixel: one 8-byte store each.  The matrix cores are used because they
//     are free (16 MFMAs per 16 KiB read), not because the problem needs them.
//   thin K (K <= 8 inputs):  a lane owns 8 consecutive output channels of a pixel (one 16-byte store), its 8 x 8 weights
//     live in registers, the pixel's 16 input bytes are one broadcast load: 64 FMAs per 16 bytes written.
//
// Both take shared or per-sample weights (grid.y = sample) in the image layout of the other forward kernels
// ([N][taps = 1][Ck], K-contiguous) and the bias / residual-merge epilogues their callers use.
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32v4 __attribute__((ext_vector_type(4)));

struct ThinParams {
    long long M;                                   // pixels (per sample when grid.y counts samples, else of the whole batch)
    int Cx, Ck, N, ldy;
    long long x_bstride, w_bstride, y_bstride;     // elements per grid.y step (0 for shared weights)
    int groups_per_wave;
    const float* bias;                             // [N] or NULL
    const void* residual;                          // thin K: y = (conv + residual) * res_gain (NULL: off)
    long long res_bstride;
    int res_ld;
    float res_gain;
};

// ---- N <= 8.  KC = Ck / 32.
template <int KC>
__global__ __launch_bounds__(256) void conv_thin_n_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, ThinParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lr = lane & 15, lh = lane >> 4;
    const long long s = blockIdx.y;
    const bf16_t* xb = x + s * p.x_bstride;
    const bf16_t* wb = w + s * p.w_bstride;
    bf16_t* yb = y + s * p.y_bstride;
    // A operand: row lr of the (zero-padded) 16 x Ck weight matrix, 8 channels at slot lh of every 32-channel step
    bf16v8 wf[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (lr < p.N) raw = *reinterpret_cast<const u32x4*>(wb + (long long)lr * p.Ck + kc * 32 + lh * 8);
        wf[kc] = __builtin_bit_cast(bf16v8, raw);
    }
    // D[n][pixel]: this lane holds n = 4 lh + e of pixel lr
    f32v4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = 4 * lh + e < p.N ? p.bias[4 * lh + e] : 0.f;
    }
    const long long n_groups = (p.M + 15) / 16;
    const long long g0 = ((long long)blockIdx.x * 4 + wid) * p.groups_per_wave;
    for (int t = 0; t < p.groups_per_wave; ++t) {
        const long long g = g0 + t;
        if (g >= n_groups) break;                                    // (wave-uniform)
        const long long pix = g * 16 + lr;
        const bool ok = pix < p.M;
        const bf16_t* xp = xb + (ok ? pix : p.M - 1) * p.Cx + lh * 8;
        u32x4 xr[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) xr[kc] = *reinterpret_cast<const u32x4*>(xp + kc * 32);
        f32v4 acc = bv;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc], __builtin_bit_cast(bf16v8, xr[kc]), acc, 0, 0, 0);
        if (ok && 4 * lh < p.ldy && lh < 2) {
            uint2 pk;
            pk.x = (unsigned)f2bf(acc[0]) | ((unsigned)f2bf(acc[1]) << 16);
            pk.y = (unsigned)f2bf(acc[2]) | ((unsigned)f2bf(acc[3]) << 16);
            *reinterpret_cast<uint2*>(yb + pix * p.ldy + 4 * lh) = pk;
        }
    }
}

// ---- K <= 8 (Cx == 8: one 16-byte vector per pixel).  NV = N / 8 lanes per pixel, 64 % NV == 0.
template <int NV, int U>                                             // U pixels in flight per lane
__global__ __launch_bounds__(256) void conv_thin_k_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, ThinParams p) {
    constexpr int PPW = 64 / NV;                                     // pixels per wave and step
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int v = lane % NV, psub = lane / NV;
    const long long s = blockIdx.y;
    const bf16_t* xb = x + s * p.x_bstride;
    const bf16_t* wb = w + s * p.w_bstride;
    bf16_t* yb = y + s * p.y_bstride;
    const bf16_t* rb = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + s * p.res_bstride : nullptr;
    float wr[8][8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(wb + (long long)(v * 8 + j) * p.Ck);
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[j][k] = bf2f((bf16_t)(raw[k >> 1] >> (16 * (k & 1))));
        bv[j] = p.bias ? p.bias[v * 8 + j] : 0.f;
    }
    const long long n_groups = (p.M + PPW - 1) / PPW;
    const long long g0 = ((long long)blockIdx.x * 4 + wid) * p.groups_per_wave;
    for (int t = 0; t < p.groups_per_wave; t += U) {
        u32x4 xr[U], rr[U];
        long long pix[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pix[u] = (g0 + t + u) * PPW + psub;
            ok[u] = t + u < p.groups_per_wave && g0 + t + u < n_groups && pix[u] < p.M;
            const long long pp = ok[u] ? pix[u] : 0;
            xr[u] = *reinterpret_cast<const u32x4*>(xb + pp * 8);
            if (rb) rr[u] = *reinterpret_cast<const u32x4*>(rb + pp * p.res_ld + v * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float xv[8], acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = bf2f((bf16_t)(xr[u][k >> 1] >> (16 * (k & 1))));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = bv[j];
#pragma unroll
                for (int k = 0; k < 8; ++k) a = fmaf(xv[k], wr[j][k], a);
                acc[j] = a;
            }
            if (rb) {
                // the residual merge of the other forward kernels (msg_common.h): applied to the ROUNDED conv result
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[j] = (bf2f(f2bf(acc[j])) + bf2f((bf16_t)(rr[u][j >> 1] >> (16 * (j & 1))))) * p.res_gain;
            }
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (unsigned)f2bf(acc[2 * j]) | ((unsigned)f2bf(acc[2 * j + 1]) << 16);
            if (ok[u]) *reinterpret_cast<u32x4*>(yb + pix[u] * p.ldy + v * 8) = o;
        }
    }
}

// ---- K <= 8 again, with the pixel vectors fetched by SCALAR loads.  The kernel above waits for its (tiny) input loads with
// s_waitcnt vmcnt, and vmcnt retires loads AND stores in order: every wait for the next pixels also waits for the previous
// pixels' stores, so the store stream -- all this kernel is -- drains once per step (3.5 TB/s against a 6.9 TB/s fill).  A wave's
// 64 / NV pixels are consecutive, i.e. one wave-uniform run of 16-byte vectors: fetched with s_load (its own counter,
// lgkmcnt), one step ahead, they never touch vmcnt, and the stores stream.  Needs M % (64 / NV) == 0 (whole runs).
template <int NV>
__global__ __launch_bounds__(256) void conv_thin_k_sload_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                bf16_t* __restrict__ y, ThinParams p) {
    constexpr int PPW = 64 / NV;                                     // pixels per wave and step
    typedef unsigned int urun __attribute__((ext_vector_type(4 * PPW)));
    typedef const __attribute__((address_space(4))) urun* crun_t;    // constant address space + uniform address = s_load
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int v = lane % NV, psub = lane / NV;
    const long long s = blockIdx.y;
    const bf16_t* xb = x + s * p.x_bstride;
    const bf16_t* wb = w + s * p.w_bstride;
    bf16_t* yb = y + s * p.y_bstride;
    const bf16_t* rb = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + s * p.res_bstride : nullptr;
    float wr[8][8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(wb + (long long)(v * 8 + j) * p.Ck);
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[j][k] = bf2f((bf16_t)(raw[k >> 1] >> (16 * (k & 1))));
        bv[j] = p.bias ? p.bias[v * 8 + j] : 0.f;
    }
    const long long n_groups = p.M / PPW;
    const long long g0 = ((long long)blockIdx.x * 4 + wid) * p.groups_per_wave;
    long long g1 = g0 + p.groups_per_wave;
    if (g1 > n_groups) g1 = n_groups;
    if (g0 >= g1) return;
    auto fetch = [&](long long g) __attribute__((always_inline)) {
        return *(crun_t)(reinterpret_cast<const char*>(xb) + g * (PPW * 16));
    };
    urun cur = fetch(g0);
    urun nx1 = fetch(g0 + 1 < g1 ? g0 + 1 : g0);
    u32x4 rcur = {0u, 0u, 0u, 0u};
    if (rb) rcur = *reinterpret_cast<const u32x4*>(rb + (g0 * PPW + psub) * p.res_ld + v * 8);
    for (long long g = g0; g < g1; ++g) {
        const long long gn = g + 1 < g1 ? g + 1 : g;
        const urun nxt = fetch(g + 2 < g1 ? g + 2 : g);               // the pixels of the step after next: lgkmcnt, not vmcnt
        u32x4 rnxt = rcur;
        if (rb) rnxt = *reinterpret_cast<const u32x4*>(rb + (gn * PPW + psub) * p.res_ld + v * 8);
        // this lane's pixel out of the run
        unsigned int xw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned int sel = cur[q];
#pragma unroll
            for (int pp = 1; pp < PPW; ++pp) sel = psub == pp ? cur[4 * pp + q] : sel;
            xw[q] = sel;
        }
        float xv[8], acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = bf2f((bf16_t)(xw[k >> 1] >> (16 * (k & 1))));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = bv[j];
#pragma unroll
            for (int k = 0; k < 8; ++k) a = fmaf(xv[k], wr[j][k], a);
            acc[j] = a;
        }
        if (rb) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = (bf2f(f2bf(acc[j])) + bf2f((bf16_t)(rcur[j >> 1] >> (16 * (j & 1))))) * p.res_gain;
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (unsigned)f2bf(acc[2 * j]) | ((unsigned)f2bf(acc[2 * j + 1]) << 16);
        *reinterpret_cast<u32x4*>(yb + (g * PPW + psub) * p.ldy + v * 8) = o;
        cur = nx1;
        nx1 = nxt;
        rcur = rnxt;
    }
}

static int thin_enabled() {
    static const int v = msg_tunable("MSG_CONV_THIN", 1);
    return v;
}

// 1: thin N, 2: thin K, 0: neither.  (1x1, stride 1, no padding, no zero insertion is implied by kh = kw = 1 and equal maps.)
extern "C" int msg_conv2d_fprop_thin_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                              int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                              int act_mode) {
    if (!thin_enabled() || kh != 1 || kw != 1 || stride != 1 || pad != 0 || in_up != 1 || pixel_shuffle || IH != OH ||
        IW != OW || B <= 0)
        return 0;
    const long long m = (long long)B * OH * OW;
    if (m < 4096) return 0;                                          // (tiny maps: launch-bound either way)
    if (N <= 8 && act_mode == 0 && ldy == 8 && Ck % 32 == 0 && Ck <= 512 && Cx >= Ck && Cx % 8 == 0) return 1;
    if (Cx == 8 && (act_mode == 0 || act_mode == 2) && N % 8 == 0 && N >= 64 && N <= 512 && 64 % (N / 8) == 0 && ldy % 8 == 0 &&
        Ck >= 8)
        return 2;
    return 0;
}

extern "C" int msg_conv2d_fprop_thin_try(const void* x, const void* w, const float* bias, void* y,
                                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                         int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                         long long w_batch_stride, const ActEpilogue* act, void* stream) {
    const int mode = msg_conv2d_fprop_thin_eligible(B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, in_up,
                                                    pixel_shuffle, act ? act->enabled : 0);
    if (!mode) return 0;
    const bool per_sample = w_batch_stride != 0;
    ThinParams p{};
    p.M = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    p.Cx = Cx; p.Ck = Ck; p.N = N; p.ldy = ldy;
    p.x_bstride = per_sample ? (long long)IH * IW * Cx : 0;
    p.w_bstride = w_batch_stride;
    p.y_bstride = per_sample ? (long long)OH * OW * ldy : 0;
    p.bias = bias;
    if (act && act->enabled == 2) {
        p.residual = act->residual; p.res_ld = act->res_ld; p.res_gain = act->res_gain;
        p.res_bstride = per_sample ? (long long)OH * OW * act->res_ld : 0;
    }
    const int samples = per_sample ? B : 1;
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* xp = (const bf16_t*)x;
    const bf16_t* wp = (const bf16_t*)w;
    bf16_t* yp = (bf16_t*)y;
    if (mode == 1) {
        const long long groups = (p.M + 15) / 16;
        // enough waves to fill the chip a few times over, each amortising its weight registers over several pixel groups
        p.groups_per_wave = (int)((groups * samples + 8191) / 8192);
        if (p.groups_per_wave < 1) p.groups_per_wave = 1;
        if (p.groups_per_wave > 8) p.groups_per_wave = 8;
        const long long blocks = (groups + 4ll * p.groups_per_wave - 1) / (4ll * p.groups_per_wave);
        if (blocks >= (1ll << 31) || samples > 65535) return 0;
        dim3 grid((unsigned)blocks, samples);
        switch (Ck / 32) {
#define THIN_N(KC_) case KC_: hipLaunchKernelGGL((conv_thin_n_kernel<KC_>), grid, dim3(256), 0, s, xp, wp, yp, p); break
            THIN_N(2); THIN_N(4); THIN_N(6); THIN_N(8); THIN_N(12); THIN_N(16);
#undef THIN_N
            default: return 0;
        }
        return 1;
    }
    const int nv = N / 8, ppw = 64 / nv;
    const long long groups = (p.M + ppw - 1) / ppw;
    p.groups_per_wave = (int)((groups * samples + 16383) / 16384);
    constexpr int U = 8;                                             // (8 pixels in flight per lane: 307 / 159 / 152 us against 321 / 171 / 158 with 4)
    p.groups_per_wave = (p.groups_per_wave + U - 1) / U * U;         // (whole steps of U pixels in flight)
    if (p.groups_per_wave < U) p.groups_per_wave = U;
    if (p.groups_per_wave > 32) p.groups_per_wave = 32;
    const long long blocks = (groups + 4ll * p.groups_per_wave - 1) / (4ll * p.groups_per_wave);
    if (blocks >= (1ll << 31) || samples > 65535) return 0;
    dim3 grid((unsigned)blocks, samples);
    static const int sload = msg_tunable("MSG_THIN_SLOAD", 1);
    // (N = 512 only: 314 -> 238 us on 6 -> 512 @256^2, B=16; with 4 pixels per wave -- N = 128 -- it measured SLOWER, 158 -> 198 us)
    if (sload && nv == 64 && p.M % ppw == 0 && (((uintptr_t)x) & 63u) == 0 && (p.x_bstride * 2) % 64 == 0) {
        switch (nv) {
#define THIN_KS(NV_) case NV_: hipLaunchKernelGGL((conv_thin_k_sload_kernel<NV_>), grid, dim3(256), 0, s, xp, wp, yp, p); return 1
            THIN_KS(8); THIN_KS(16); THIN_KS(32); THIN_KS(64);
#undef THIN_KS
            default: return 0;
        }
    }
    switch (nv) {
#define THIN_K(NV_) case NV_: hipLaunchKernelGGL((conv_thin_k_kernel<NV_, U>), grid, dim3(256), 0, s, xp, wp, yp, p); break
        THIN_K(8); THIN_K(16); THIN_K(32); THIN_K(64);
#undef THIN_K
        default: return 0;
    }
    return 1;
}

