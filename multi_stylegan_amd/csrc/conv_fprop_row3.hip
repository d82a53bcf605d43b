// a3/a4: 3x3 'same' convolution (forward and data gradient) on maps at least 64 pixels wide, with the ACTIVATION tile
// shared by the three horizontal taps (bf16; the 512-channel 64^2 .. 256^2 layers of the generator and the wide
// discriminator layers -- where most of the forward / data-gradient time is).
//
// conv_fprop_pp.hip stages, for every K-tile (one tap x 64 channels), 256 activation rows and 256 weight rows; the
// global -> LDS staging is what bounds that kernel ("where the conv kernels stand").  The three horizontal
// taps of a kernel row read the SAME activation pixels shifted by one, so here the K loop runs (kh, channel chunk, kw)
// with kw innermost: the activation tile of a (kh, chunk) -- the 256 pixels of the output tile plus one neighbour on each
// side of every image-row segment, zeros at the image border -- is staged ONCE and the fragment reads of tap kw start
// kw rows further down; only the 256 weight rows change per K-step.  Staging per K-step: 32 + 33/3 = 43 KiB instead of
// 64.  (conv_wgrad_row3.hip does the same for the weight gradient.)
//
// Structure: 256 x 256 output tile, four waves with 128 x 128 wave tiles (256 accumulators in the unified VGPR/AGPR
// file, one workgroup per CU), LDS-DMA staging through buffer descriptors, the K-step software-pipelined inside the wave
// (fragment reads of sub-step kk+1 and the DMA of the next K-step / next activation tile between the MFMAs of sub-step
// kk), one workgroup barrier per K-step -- the one-wave-per-SIMD design measured in round 1.  LDS: 2 activation buffers of 264 rows +
// 2 weight buffers of 256 rows, 128-B rows with XOR-swizzled 16-B slots (130 KiB); the epilogue (transposed
// accumulators, packed LDS writes, batched 16-B stores, optional fused activation / residual merge) reuses it.
// An output tile is 256 consecutive pixels = one or more whole image-row segments (map width 64, 128, or a multiple of
// 256); segment s occupies LDS rows s*(len+2) .. s*(len+2)+len+1, i.e. output pixel p of segment s, tap kw, reads row
// p + kw + 2 s.
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* lds_t;

struct ConvParamsR3 {
    int B, IH, IW, Cx, Ck, OH, OW, N, ldy;
    int per_sample, seg_len, n_seg;
    long long x_bstride, w_bstride, y_bstride;     // elements
    int Mtot, n_chunks, n_iters, m_tiles, n_tiles;
    ActEpilogue act;
};

constexpr int HROW = 128;
constexpr int H_OOB = (int)0x80000000;

// MI / NCOLB = 32-row / 32-column blocks per wave (waves are 2 x 2): <4,4> the 256 x 256 tile with 128 x 128 wave tiles, one
// workgroup per CU; <2,2> a 128 x 128 tile with 64 x 64 wave tiles and TWO workgroups per CU for layers with 128 / 384
// output channels (<4,2>, 256 x 128 with one workgroup per CU, measured no better than the plain 128x128 kernel).
template <int MI, int NCOLB>
__global__ __launch_bounds__(256, MI == 2 ? 2 : 1) void conv_fprop_row3_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                 bf16_t* __restrict__ y, const float* __restrict__ bias,
                                                                 ConvParamsR3 p) {
    constexpr int VEC = 8, BKE = 64, ESZ = 2;
    constexpr int HN = 64 * NCOLB, HB = HN * HROW, WN = 32 * NCOLB;      // tile columns, weight buffer bytes, wave-tile columns
    constexpr int NBP = 2 * NCOLB;                                    // weight pieces (8 rows) per wave and K-step
    constexpr int HM = 64 * MI, WM = 32 * MI;                         // tile rows, wave-tile rows
    constexpr int NAP = 2 * MI;                                       // activation pieces per wave and group (+ piece NAP: rows HM..HM+7, wave 0)
    constexpr int HA = (HM + 8) * HROW;                               // activation buffer
    __shared__ __attribute__((aligned(16))) char smem[2 * HA + 2 * HB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid_u >> 1, wn = wid_u & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (int)(L % p.n_tiles) * HN;
    const int m0 = (int)(L / p.n_tiles) * HM;
    const int bz = blockIdx.z;
    const int ohw = p.OH * p.OW;
    const int seg = p.seg_len;

    // ---- staging (LDS-DMA through buffer loads): one wave-instruction fills 1 KiB = 8 consecutive rows in lane order.
    // Weights: wave w moves rows 64 w + 8 j + (lane >> 3), j = 0..7, of every K-step.  Activations: the same rows of the
    // 264-row buffer once per (kh, chunk), plus rows 256..263 (piece 8, wave 0 only).  lane & 7 is the PHYSICAL slot;
    // the lane fetches the logical slot that the swizzle puts there.  Addressing = descriptor + per-lane 32-bit offset
    // (recomputed per kernel row for the activations; out-of-image rows get an out-of-range offset, i.e. zeros) + SGPR
    // offset (the K position).
    const int slot_phys = lane & 7;
    const char* xb = (const char*)x + (p.per_sample ? (long long)bz * p.x_bstride * ESZ : 0);
    const char* wb = (const char*)w + (p.per_sample ? (long long)bz * p.w_bstride * ESZ : 0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, 0x7ffffff0, 0x00020000);
    int a_ih0[NAP + 1], a_b32[NAP + 1], a_sl[NAP + 1], va[NAP + 1], vb[NBP];
    unsigned a_okmask = 0;
#pragma unroll
    for (int j = 0; j < NAP + 1; ++j) {
        const int row = (j < NAP ? wid * (HM / 4) + 8 * j : HM) + (lane >> 3);    // row of the activation buffer
        const int sl = slot_phys ^ ((row >> 1) & 7);
        const int s = row / (seg + 2), pos = row - s * (seg + 2);                 // segment, position (0 and seg+1: halo)
        const int inner = min(max(pos - 1, 0), seg - 1);
        const int m = m0 + s * seg + inner;                                       // the pixel (or the halo's neighbour)
        bool ok = (s < p.n_seg) & (m < p.Mtot) & (j < NAP || wid == 0);
        const int mm = ok ? m : 0;
        const int b = p.per_sample ? 0 : mm / ohw;
        const int pix = p.per_sample ? mm : mm - b * ohw;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        const int iw = ow + (pos == 0 ? -1 : (pos == seg + 1 ? 1 : 0));
        ok = ok & ((unsigned)iw < (unsigned)p.IW);
        a_okmask |= (ok ? 1u : 0u) << j;
        a_ih0[j] = oh - 1;
        a_sl[j] = sl;
        // offset of kernel row 0 (may be "negative" for the top image row: only used when that row is in range)
        a_b32[j] = (int)(((long long)b * p.x_bstride + sl * VEC) * ESZ) + ((oh - 1) * p.IW + iw) * p.Cx * ESZ;
        va[j] = H_OOB;
        if (j < NBP) {
            const int wrow = wid * (WN / 2) + 8 * j + (lane >> 3);
            const int n = n0 + wrow;
            const int slb = slot_phys ^ ((wrow >> 1) & 7);
            vb[j] = n < p.N ? (int)(((long long)n * 9 * p.Ck + slb * VEC) * ESZ) : H_OOB;
        }
    }
    const bool ragged = (p.Cx % BKE) != 0;
    auto set_kh = [&](int kh) __attribute__((always_inline)) {                    // activation offsets of kernel row kh
        const int tap_off = kh * p.IW * p.Cx * ESZ;
#pragma unroll
        for (int j = 0; j < NAP + 1; ++j) {
            const bool ok = ((a_okmask >> j) & 1u) & ((unsigned)(a_ih0[j] + kh) < (unsigned)p.IH);
            va[j] = ok ? a_b32[j] + tap_off : H_OOB;
        }
    };
    // activation piece j of (kh set by set_kh, chunk) into activation buffer `abuf`
    auto dma_a = [&](int j, int abuf, int chunk, bool live) __attribute__((always_inline)) {
        if (j == NAP && wid_u != 0) return;
        const bool a_zero = !live | (ragged & (chunk * BKE + a_sl[j] * VEC + VEC > p.Cx));
        lds_t la = (lds_t)(smem + abuf * HA + (j < NAP ? wid_u * (HM / 4) + 8 * j : HM) * HROW);
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass instantiates this template too: it must not see the device builtin)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, la, 16, a_zero ? H_OOB : va[j], chunk * HROW, 0, 0);
#endif
    };
    // weight piece j of K-step (tap, chunk) into weight buffer `bbuf`
    auto dma_b = [&](int j, int bbuf, int tap, int chunk, bool live) __attribute__((always_inline)) {
        lds_t la = (lds_t)(smem + 2 * HA + bbuf * HB + (wid_u * (WN / 2) + 8 * j) * HROW);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, la, 16, live ? vb[j] : H_OOB, (tap * p.n_chunks + chunk) * HROW, 0, 0);
#endif
    };

    f32x16 acc[MI][NCOLB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NCOLB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- prologue: activation tile of group 0 (kernel row 0, chunk 0), weights of K-step 0
    set_kh(0);
#pragma unroll
    for (int j = 0; j < NAP + 1; ++j) dma_a(j, 0, 0, true);
#pragma unroll
    for (int j = 0; j < NBP; ++j) dma_b(j, 0, 0, 0, true);

    // cursors: current K-step = (kh, chunk, kw); the group being loaded = (kh_l, chunk_l)
    int kh = 0, chunk = 0, kw = 0;
    int kh_l = 0, chunk_l = 0;
    // fragment addressing: activation row of output row (wm*128 + i*32 + lr), tap kw: + kw + 2 * segment
    int seg_of[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) seg_of[i] = (wm * WM + i * 32) / seg;
    const int sxb = (lr >> 1) & 7;
    const int fb_base = 2 * HA + (wn * WN + lr) * HROW;

    for (int it = 0; it < p.n_iters; ++it) {
        // This wave's LDS-DMA for step `it` must have LANDED before the barrier: say so explicitly.  (The compiler's own
        // wait in front of __syncthreads() is derived from alias analysis of the DMA destinations and was vmcnt(2) here --
        // the last two pieces could still be in flight: a race that showed up as sporadically wrong output channels.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // all reads of the other buffers are done
        const bool more = it + 1 < p.n_iters;
        // next K-step's coordinates
        int kw_n = kw + 1, chunk_n = chunk, kh_n = kh;
        if (kw_n == 3) { kw_n = 0; if (++chunk_n == p.n_chunks) { chunk_n = 0; ++kh_n; } }
        const int grp = it / 3;                                  // (uniform; it / 3 by multiply-shift)
        const int abuf = grp & 1, bbuf = it & 1;
        // at the first step of a group the NEXT group's activation tile starts loading
        bool more_a = false;
        if (kw == 0) {
            chunk_l = chunk + 1; kh_l = kh;
            if (chunk_l == p.n_chunks) { chunk_l = 0; ++kh_l; }
            if (kh_l != kh && kh_l < 3) set_kh(kh_l);
        }
        more_a = kh_l < 3;
        const char* sA = smem + abuf * HA;
        const char* sB = smem + bbuf * HB;
        int a_row[MI], a_sx[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = wm * WM + i * 32 + lr + kw + 2 * seg_of[i];
            a_row[i] = r * HROW;
            a_sx[i] = (r >> 1) & 7;
        }
        bf16v8 fa[2][MI], fb[2][NCOLB];
#pragma unroll
        for (int t = 0; t < MI; ++t) fa[0][t] = *reinterpret_cast<const bf16v8*>(sA + a_row[t] + ((lh ^ a_sx[t]) << 4));
#pragma unroll
        for (int t = 0; t < NCOLB; ++t)
            fb[0][t] = *reinterpret_cast<const bf16v8*>(sB + fb_base + t * 32 * HROW + ((lh ^ sxb) << 4));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NCOLB; ++j)     // operands swapped: transposed accumulators (see the epilogue)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
                if (kk + 1 < 4) {
                    const int s2 = 2 * (kk + 1) + lh;
                    fa[(kk + 1) & 1][i] = *reinterpret_cast<const bf16v8*>(sA + a_row[i] + ((s2 ^ a_sx[i]) << 4));
                    if (i < NCOLB) fb[(kk + 1) & 1][i] = *reinterpret_cast<const bf16v8*>(sB + fb_base + i * 32 * HROW + ((s2 ^ sxb) << 4));
                }
                // Staging instructions of the K-step -- NBP weight pieces of the next K-step, then (first K-step of a group)
                // the activation pieces of the next group -- two per group of NCOLB MFMAs, from the START of the step: the earlier they
                // are issued the more time the LDS-DMA has to land before the barrier.  (Measured on 3x3 512->512 @256^2:
                // spread over the whole step 4290 us, one per group from the start 4180, two per group 4040, more: no gain.)
                const int slot = kk * MI + i;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int d = 2 * slot + q;
                    if (d < NBP) dma_b(d, bbuf ^ 1, kh_n * 3 + kw_n, chunk_n, more);
                    else if (d < NBP + NAP + 1) {
                        // ALL activation pieces of the next group go out in the group's FIRST K-step: they come from
                        // HBM rather than L2 and need the two remaining K-steps to land (spread three per K-step, the
                        // last three had one K-step).  Small gain: 3x3 128->128 @256^2 476 -> 452 us, 512->512 @256^2 ~1 %
                        if (kw == 0) dma_a(d - NBP, abuf ^ 1, chunk_l, more_a);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        kw = kw_n; chunk = chunk_n; kh = kh_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the dummy pieces of the last step write zeros: they must land first)
    __syncthreads();

    // ---- epilogue: wave-private WM x WN bf16 patch in LDS, then 16-B stores.  Transposed accumulators: lane (lr, lh)
    // owns pixel 32 i + lr and, per group g = e >> 2, the four CONSECUTIVE channels 32 j + 8 g + 4 lh + (0..3): one packed
    // 8-byte LDS write each; 8-byte unit u of row r lives at unit u ^ (r & 15).
    constexpr int PITCH = WN * ESZ;
    char* ep = smem + wid * (WM * PITCH);
#pragma unroll
    for (int j = 0; j < NCOLB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int unit = 8 * j + 2 * g + lh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
                const int nb = n0 + wn * WN + 4 * unit;
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = nb + e < p.N ? bias[nb + e] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = i * 32 + lr;
                uint2 pk;
                pk.x = (unsigned)f2bf(acc[i][j][4 * g + 0] + bv[0]) | ((unsigned)f2bf(acc[i][j][4 * g + 1] + bv[1]) << 16);
                pk.y = (unsigned)f2bf(acc[i][j][4 * g + 2] + bv[2]) | ((unsigned)f2bf(acc[i][j][4 * g + 3] + bv[3]) << 16);
                *reinterpret_cast<uint2*>(ep + row * PITCH + ((unit ^ (row & 15)) << 3)) = pk;
            }
        }
    constexpr int LPR = WN / VEC;                               // lanes per patch row: 16 / 8
    constexpr int RPP = 64 / LPR;                               // rows per pass: 4 / 8
    constexpr int NP = 64 / RPP;                                // passes per half patch (64 rows): 16 / 8
    const int er = lane / LPR, u16 = lane % LPR, ec = u16 * VEC;
    const int n = n0 + wn * WN + ec;
    const bool n_ok = n < p.N;
    const int lim = p.N - n;
    // half-patches of 64 rows: row coordinates stepped, all rows / residual vectors of a half requested before any is
    // used
#pragma unroll 1
    for (int half = 0; half < MI / 2; ++half) {
        int gp[NP];
        float a_bias[VEC], a_noise[NP];
        {
            const int m_first = min(m0 + wm * WM + half * 64 + er, p.Mtot - 1);
            int b = p.per_sample ? bz : m_first / ohw;
            const int pix0 = p.per_sample ? m_first : m_first - b * ohw;
            int oh = pix0 / p.OW, ow = pix0 - oh * p.OW;
            const bool want_noise = p.act.enabled == 1 && p.act.noise;
            const float nw = want_noise ? p.act.noise_w[0] : 0.f;
            if (p.act.enabled == 1) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) a_bias[e] = (p.act.bias && n + e < p.N) ? p.act.bias[n + e] : 0.f;
            }
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const bool ok = m0 + wm * WM + half * 64 + pass * RPP + er < p.Mtot;
                const int pix = oh * p.OW + ow;
                gp[pass] = ok ? b * ohw + pix : -1;
                a_noise[pass] = (want_noise && ok) ? nw * p.act.noise[(long long)(p.act.noise_batch == 1 ? 0 : b) * ohw + pix] : 0.f;
                ow += RPP;
                while (ow >= p.OW) { ow -= p.OW; ++oh; }
                if (!p.per_sample) while (oh >= p.OH) { oh -= p.OH; ++b; }
            }
        }
        if (half == 0) __syncthreads();
        u32x4 v[NP];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int row = half * 64 + pass * RPP + er;
            v[pass] = *reinterpret_cast<const u32x4*>(ep + row * PITCH + ((u16 ^ ((row & 15) >> 1)) << 4));
        }
        if (er & 1) {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) v[pass] = u32x4{v[pass][2], v[pass][3], v[pass][0], v[pass][1]};
        }
        if (p.act.enabled == 1) {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) v[pass] = act_epilogue_apply<bf16_t>(v[pass], a_bias, a_noise[pass], p.act.alpha, p.act.scale);
        } else if (p.act.enabled == 2) {
            const bf16_t* rbase = reinterpret_cast<const bf16_t*>(p.act.residual) + (n_ok ? n : 0);
            u32x4 r[NP];
#pragma unroll
            for (int pass = 0; pass < NP; ++pass)
                r[pass] = *reinterpret_cast<const u32x4*>(rbase + (long long)max(gp[pass], 0) * p.act.res_ld);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) v[pass] = residual_epilogue_apply<bf16_t>(v[pass], r[pass], p.act.res_gain);
        }
        bf16_t* ybase = y + (p.per_sample ? (long long)bz * p.y_bstride : 0) + n;
        if (n_ok && lim >= VEC) {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const long long g = p.per_sample ? (long long)(gp[pass] - bz * ohw) : (long long)gp[pass];
                if (gp[pass] >= 0) *reinterpret_cast<u32x4*>(ybase + g * p.ldy) = v[pass];
            }
        } else if (n_ok) {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                if (gp[pass] < 0) continue;
                const long long g = p.per_sample ? (long long)(gp[pass] - bz * ohw) : (long long)gp[pass];
                bf16_t* dst = ybase + g * p.ldy;
                for (int e = 0; e < lim; ++e) dst[e] = (bf16_t)(v[pass][e >> 1] >> (16 * (e & 1)));
            }
        }
    }
}

// Tile width for N output channels: 256 columns unless that leaves a mostly empty last tile (N = 128, 384), then 128
// columns; 0 = neither fits (the 15 % padding rule of the pp kernel).
static int row3_tile_columns(int N) {
    if (N >= 256 && (long long)((N + 255) / 256) * 256 * 100 <= (long long)N * 115) return 256;
    if (N >= 128 && (long long)((N + 127) / 128) * 128 * 100 <= (long long)N * 115) return 128;
    return 0;
}

// Which problems take this kernel (shared by the launcher below and by msg_conv2d_fprop_plan): a 3x3 convolution whose
// output map equals its input map is the stride-1, pad-1, no-zero-insertion 'same' convolution.
extern "C" int msg_conv2d_fprop_row3_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                                              int kh, int kw, long long w_batch_stride) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("MSG_CONV_ROW3"); enabled = e ? atoi(e) : 1; }
    if (!enabled || kh != 3 || kw != 3 || IH != OH || IW != OW || Ck % 64) return 0;
    const bool per_sample = w_batch_stride != 0;
    const long long mtot = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    // The 128 x 128 variant (two workgroups per CU) takes the layers with 128 / 384 output channels from the plain 128x128
    // kernel: 3x3 128->128 @256^2 516 -> 476 us, 256->128 @256^2 866 -> 819, 256->384 @128^2 650 -> 614, 384->384 @64^2
    // 235 -> 205.  MSG_CONV_ROW3_NARROW=0 switches it off (A/B).
    static int narrow = -1;
    if (narrow < 0) { const char* e = getenv("MSG_CONV_ROW3_NARROW"); narrow = e ? atoi(e) : 1; }
    // 32-wide maps: the 128 x 128 tile holds four image rows (4 x 34 = 136 buffer rows) whatever the channel count:
    // 3x3 768->768 @32^2, B=16: 229 -> 179 us, 1024->768: 318 -> 272 (they ran on the ping-pong kernel).  MSG_CONV_ROW3_W32=0: off.
    static int w32 = -1;
    if (w32 < 0) { const char* e = getenv("MSG_CONV_ROW3_W32"); w32 = e ? atoi(e) : 1; }
    int hn = row3_tile_columns(N);
    if (OW == 32 && w32 && N >= 128 && (long long)((N + 127) / 128) * 128 * 100 <= (long long)N * 115) hn = 128;
    if (!hn || (hn == 128 && !narrow)) return 0;
    const int hm = hn;                             // square tiles: 256 x 256 or 128 x 128
    if (!((OW >= 64 && hm % OW == 0) || OW % hm == 0 || (OW == 32 && hm == 128))) return 0;   // whole image-row segments per tile, <= 8 halo rows
    if (mtot < 1024 || mtot >= (1ll << 31) || mtot % hm) return 0;
    const long long x_bytes = (long long)(per_sample ? 1 : B) * IH * IW * Cx * 2;
    const long long w_bytes = (long long)N * 9 * Ck * 2;
    if (x_bytes >= 0x7ffffff0ll || w_bytes >= 0x7ffffff0ll) return 0;                   // 31-bit buffer offsets
    if ((long long)9 * (Ck / 64) * HROW >= (1ll << 24)) return 0;
    const long long blocks = (mtot / hm) * ((N + hn - 1) / hn);
    if (blocks * (per_sample ? B : 1) < (hm == 256 ? 224 : 448) || blocks >= (1ll << 31)) return 0;
    return hm == 256 ? 1 : 2;                      // 1: 256 x 256 tile, 2: 128 x 128 tile
}

// Called by msg_conv2d_fprop (conv_fprop.hip) before the other large-tile kernels; returns 1 if it launched.
extern "C" int msg_conv2d_fprop_row3_try(const void* x, const void* w, const float* bias, void* y,
                                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                         int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                         long long w_batch_stride, const ActEpilogue* act, void* stream) {
    if (stride != 1 || pad != 1 || in_up != 1 || pixel_shuffle ||
        !msg_conv2d_fprop_row3_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride))
        return 0;
    const bool per_sample = w_batch_stride != 0;
    const long long mtot = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    const int n_chunks = Ck / 64;
    ConvParamsR3 p{};
    p.B = B; p.IH = IH; p.IW = IW; p.Cx = Cx; p.Ck = Ck; p.OH = OH; p.OW = OW; p.N = N; p.ldy = ldy;
    p.per_sample = per_sample;
    const int hn = msg_conv2d_fprop_row3_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride) == 1 ? 256 : 128, hm = hn;
    p.seg_len = OW < hm ? OW : hm;
    p.n_seg = hm / p.seg_len;
    if (act) p.act = *act;
    p.x_bstride = (long long)IH * IW * Cx;
    p.w_bstride = w_batch_stride;
    p.y_bstride = (long long)OH * OW * ldy;
    p.Mtot = (int)mtot;
    p.n_chunks = n_chunks;
    p.n_iters = 9 * n_chunks;
    p.m_tiles = (int)(mtot / hm);
    p.n_tiles = (N + hn - 1) / hn;
    const long long blocks = (long long)p.m_tiles * p.n_tiles;
    dim3 grid((unsigned)blocks, 1, per_sample ? B : 1);
    if (hn == 256)
        hipLaunchKernelGGL((conv_fprop_row3_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    else
        hipLaunchKernelGGL((conv_fprop_row3_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    return 1;
}
