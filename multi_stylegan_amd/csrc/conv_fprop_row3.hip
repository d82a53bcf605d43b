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
// Structure: 256 x 256 output tile, four waves with 128 x 128 wave tiles (256 accumulators in the AGPR half of the
// unified register file, one workgroup per CU), v_mfma_f32_16x16x32_bf16, LDS-DMA staging through buffer descriptors.
// The K loop is software-pipelined inside the wave and scheduled by hand: one barrier per K-step placed in front of the
// step's last sub-step, fragment reads / staging instructions / address arithmetic spread one or two per MFMA gap, the
// four waves' staging instructions interleaved in time (per-wave instances of the loop), the three horizontal taps
// unrolled (no branches in the loop); see the comments at the loop.  LDS: 2 activation buffers of 264 rows +
// 2 weight buffers of 256 rows, 128-B rows with XOR-swizzled 16-B slots (130 KiB); the epilogue (transposed
// accumulators, packed LDS writes, batched 16-B stores, optional fused activation / residual merge) reuses it.
// An output tile is 256 consecutive pixels = one or more whole image-row segments (map width 64, 128, or a multiple of
// 256); segment s occupies LDS rows s*(len+2) .. s*(len+2)+len+1, i.e. output pixel p of segment s, tap kw, reads row
// p + kw + 2 s.
#include "msg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* lds_t;

struct ConvParamsR3 {
    int B, IH, IW, Cx, Ck, OH, OW, N, ldy;
    int per_sample, seg_len, n_seg;
    long long x_bstride, w_bstride, y_bstride;     // elements
    int Mtot, n_chunks, n_iters, m_tiles, n_tiles;
    int seg_magic;                 // row / (seg_len + 2) == (row * seg_magic) >> 16 for every row of the activation buffer
    int ow_shift, ohw_shift;       // log2(OW), log2(OH * OW) when both are powers of two, else -1 (generic divisions)
    ActEpilogue act;
};

constexpr int HROW = 128;
constexpr int H_OOB = (int)0x80000000;

#ifdef MSG_ROW3_STAMPS
// diagnostic build only (tools/row3_stamps.py; never ship or benchmark it): cycle stamps of K-steps 9..11 of every wave
// of the LAST 256 workgroups of the launch (the first 256 start in lock-step at the boost clock and fight over the same weight
// rows: their waits are not the steady state's)
__device__ unsigned long long g_row3_stamps[256 * 4 * 3 * 8];
#define R3_STAMP(k) do { if (it >= 9 && it < 12 && L + 256 >= gridDim.x && blockIdx.z == gridDim.z - 1 && lane == 0) { unsigned long long tt; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory"); \
    g_row3_stamps[(((L + 256 - gridDim.x) * 4 + wid_u) * 3 + (it - 9)) * 8 + (k)] = tt; } } while (0)
extern "C" int msg_row3_debug_read(void* host_dst, int nbytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_row3_stamps), nbytes) == hipSuccess ? 0 : -1;
}
// in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles (s_memtime) and the 100 MHz reference counter
// (s_memrealtime) at kernel entry (0), at the start (1) and the end (2) of the K loop and at kernel exit (3) of every wave
__device__ unsigned long long g_row3_clock[256 * 4 * 8];
// (the LAST 256 workgroups of the last sample: the clock the chip holds deep into the launch, not the boost of its first tiles)
#define R3_CLOCK(k) do { if (L + 256 >= gridDim.x && blockIdx.z == gridDim.z - 1 && lane == 0) { unsigned long long tc, tr; \
    const unsigned cl_ = L + 256 - gridDim.x; \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc), "=s"(tr) :: "memory"); \
    g_row3_clock[(cl_ * 4 + wid_u) * 8 + 2 * (k)] = tc; g_row3_clock[(cl_ * 4 + wid_u) * 8 + 2 * (k) + 1] = tr; } } while (0)
extern "C" int msg_row3_clock_read(void* host_dst, int nbytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_row3_clock), nbytes) == hipSuccess ? 0 : -1;
}
#else
#define R3_STAMP(k) do {} while (0)
#define R3_CLOCK(k) do {} while (0)
#endif

// MI / NCOLB = 32-row / 32-column blocks per wave (waves are 2 x 2): <4,4> the 256 x 256 tile with 128 x 128 wave tiles, one
// workgroup per CU; <2,2> a 128 x 128 tile with 64 x 64 wave tiles and TWO workgroups per CU for layers with 128 / 384
// output channels (<4,2>, 256 x 128 with one workgroup per CU, measured no better than the plain 128x128 kernel).
// EPI = 1: the epilogue is the backward of the activation in front of this conv's input (ActEpilogue::enabled == 3, see
// msg_common.h) -- its own instantiation, so that the kernels of the forward / plain data-gradient launches are the code they were.
template <int MI, int NCOLB, bool S16, int EPI = 0>
__global__ __launch_bounds__(256, MI == 2 ? 2 : 1) void conv_fprop_row3_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                 bf16_t* __restrict__ y, const float* __restrict__ bias,
                                                                 ConvParamsR3 p) {
    constexpr int VEC = 8, ESZ = 2;
    constexpr int HN = 64 * NCOLB, HB = HN * HROW, WN = 32 * NCOLB;      // tile columns, weight buffer bytes, wave-tile columns
    constexpr int NBP = 2 * NCOLB;                                    // weight pieces (8 rows) per wave and K-step
    constexpr int HM = 64 * MI, WM = 32 * MI;                         // tile rows, wave-tile rows
    constexpr int NAP = 2 * MI;                                       // activation pieces per wave and group (+ piece NAP: rows HM..HM+7, wave 0)
    constexpr int HA = (HM + 8) * HROW;                               // activation buffer
#if defined(MSG_ROW3_NO_STAGGER)
    constexpr bool STAGGER = false;
#else
    constexpr bool STAGGER = true;        // (the 128 x 128 tile too: +1..2 %)
#endif
    // (+ HM noise values and HN bias values of the fused activation stage, fetched at kernel START into LDS the K loop
    //  does not use: with one workgroup per CU nothing overlaps the epilogue, and its dependent global loads -- bias vector and
    //  per-pixel noise, twice per tile -- sat in front of every half patch with their full latency; tools/layer_probe.py:
    //  the fused stage cost 330 us of a 3 300 us launch)
    constexpr int EP_OFF = 2 * HA + 2 * HB;
    // (EPI 1: + the tile's sign bytes, [HM rows][HN / 8], fetched at kernel start like the noise: see the epilogue)
    constexpr int SIGN_OFF = EP_OFF + (HM + HN) * 4;
    __shared__ __attribute__((aligned(16))) char smem[SIGN_OFF + (EPI == 1 ? HM * (HN / 8) : 0)];
    float* const ep_noise = reinterpret_cast<float*>(smem + EP_OFF);
    float* const ep_bias = ep_noise + HM;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid_u >> 1, wn = wid_u & 1;
    // MFMA shape: v_mfma_f32_32x32x16_bf16 (a lane holds row lane & 31, 8 channels at 16-B slot lane >> 5 of a 16-channel
    // sub-step) or, S16, v_mfma_f32_16x16x32_bf16 (row lane & 15, slot lane >> 4 of a 32-channel sub-step): the same LDS
    // image, the same bytes read per K-step, and on real data the higher sustained clock
    constexpr int BLK = S16 ? 16 : 32;                                // rows / columns of an MFMA block
    constexpr int NSUB = S16 ? 2 : 4, SLOTS = S16 ? 4 : 2;            // sub-steps per K-step, 16-B slots per sub-step
    constexpr int NA = WM / BLK, NB = WN / BLK;                       // blocks of a wave tile (NA == NB)
    const int lr = lane & (BLK - 1), lh = lane / BLK;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (int)(L % p.n_tiles) * HN;
    const int m0 = (int)(L / p.n_tiles) * HM;
    const int bz = blockIdx.z;
    const int ohw = p.OH * p.OW;
    const int seg = p.seg_len;
    R3_CLOCK(0);
    if (p.act.enabled == 1 || (EPI == 1 && p.act.noise)) {
        const bool want_noise = p.act.noise != nullptr;
        const float nw = EPI == 1 ? 1.f : (want_noise ? p.act.noise_w[0] : 0.f);     // (EPI 1: the noise itself, for the noise-weight gradient)
        for (int t = tid; t < HM; t += 256) {
            const int m = m0 + t;
            const bool ok = want_noise && m < p.Mtot;
            const int mm = ok ? m : 0;
            // (pixel index over the whole batch, or -- one noise map for the batch -- inside its sample)
            long long idx = p.per_sample ? (long long)(p.act.noise_batch == 1 ? 0 : bz) * ohw + mm : (long long)mm;
            if (!p.per_sample && p.act.noise_batch == 1) idx = p.ohw_shift >= 0 ? mm & (ohw - 1) : mm % ohw;
            ep_noise[t] = ok ? nw * p.act.noise[idx] : 0.f;
        }
        if (EPI == 0)
            for (int t = tid; t < HN; t += 256) ep_bias[t] = (p.act.bias && n0 + t < p.N) ? p.act.bias[n0 + t] : 0.f;
    }
    if constexpr (EPI == 1) {
        // The sign bytes of this tile (written by ANOTHER launch in tiles of mask_tile_m x mask_tile_n, act_mask_index) into LDS
        // now, in 16-byte units = one row x 128 channels, which lie inside one tile of the producer (mask_tile_n is a multiple
        // of 128, or the whole row): fetched in the epilogue they sat in front of every half patch with their full latency,
        // +29 % on the 128-channel layers.
        if (p.act.mask) {
            constexpr int UPR = HN / 128;                                    // units per row
            const int vpt = p.act.mask_tile_n >> 3, ntn = p.N / p.act.mask_tile_n;
            for (int t = tid; t < HM * UPR; t += 256) {
                const int row = t / UPR, un = t - row * UPR;
                const long long qg = (p.per_sample ? (long long)bz * ohw : 0) + m0 + row;
                const int cvg = (n0 >> 3) + 16 * un;
                const long long trow = qg / p.act.mask_tile_m;
                const int tn = cvg / vpt;
                const unsigned char* src = p.act.mask +
                    ((trow * ntn + tn) * p.act.mask_tile_m + (qg - trow * p.act.mask_tile_m)) * vpt + (cvg - tn * vpt);
                *reinterpret_cast<u32x4*>(smem + SIGN_OFF + row * (HN / 8) + 16 * un) = *reinterpret_cast<const u32x4*>(src);
            }
        }
    }

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
    // Registers are what this kernel is short of (256 accumulators + two sets of fragments), so the staging state is
    // kept small: per activation piece ONE offset (kernel row 0) + one shared word of flag bits, per weight piece
    // nothing -- a wave's weight rows are 8 j + (lane >> 3), i.e. a uniform stride per piece (SGPR offset) and a swizzle
    // term that only alternates with the parity of j.  (Eligibility: Cx a multiple of 64, N a multiple of the tile width.)
    int a_b32[NAP + 1];
    unsigned a_flags = 0;             // bit j: piece j's row is a pixel of the map; 9 + j: ... of its top row; 18 + j: ... bottom row
    // (One wave per SIMD: every instruction of this prologue is four cycles nothing else fills -- in-kernel stamps, entry
    //  -> K loop was 14 200 cycles of a 227 000-cycle workgroup with three generic integer divisions per piece.  The
    //  segment comes from a multiply, and maps whose sides are powers of two -- all of the reference's -- take shifts.)
    auto piece_offsets = [&](auto pow2_tag) __attribute__((always_inline)) {
        constexpr bool POW2 = decltype(pow2_tag)::value;
#pragma unroll
        for (int j = 0; j < NAP + 1; ++j) {
            const int row = (j < NAP ? wid * (HM / 4) + 8 * j : HM) + (lane >> 3);    // row of the activation buffer
            const int sl = slot_phys ^ ((row >> 1) & 7);
            const int s = (row * p.seg_magic) >> 16, pos = row - s * (seg + 2);       // segment, position (0 and seg+1: halo)
            const int inner = min(max(pos - 1, 0), seg - 1);
            const int m = m0 + s * seg + inner;                                       // the pixel (or the halo's neighbour)
            bool ok = (s < p.n_seg) & (m < p.Mtot) & (j < NAP || wid == 0);
            const int mm = ok ? m : 0;
            int b, pix, oh, ow;
            if (POW2) {
                b = p.per_sample ? 0 : mm >> p.ohw_shift;
                pix = p.per_sample ? mm : mm & (ohw - 1);
                oh = pix >> p.ow_shift, ow = pix & (p.OW - 1);
            } else {
                b = p.per_sample ? 0 : mm / ohw;
                pix = p.per_sample ? mm : mm - b * ohw;
                oh = pix / p.OW, ow = pix - oh * p.OW;
            }
            const int iw = ow + (pos == 0 ? -1 : (pos == seg + 1 ? 1 : 0));
            ok = ok & ((unsigned)iw < (unsigned)p.IW);
            a_flags |= (ok ? 1u : 0u) << j | (oh == 0 ? 1u : 0u) << (9 + j) | (oh == p.IH - 1 ? 1u : 0u) << (18 + j);
            // offset of kernel row 0 (may be "negative" for the top image row: only used when that row is in range)
            a_b32[j] = (int)(((long long)b * p.x_bstride + sl * VEC) * ESZ) + ((oh - 1) * p.IW + iw) * p.Cx * ESZ;
        }
    };
    if (p.ow_shift >= 0) piece_offsets(std::true_type{});
    else piece_offsets(std::false_type{});
    static_assert(NAP + 1 <= 9, "flag bits");
    int vb2[2];                                                                   // weight offsets of pieces 0 and 1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int wrow = wid * (WN / 2) + 8 * j + (lane >> 3);
        const int slb = slot_phys ^ ((wrow >> 1) & 7);
        vb2[j] = (int)(((long long)(n0 + wrow) * 9 * p.Ck + slb * VEC) * ESZ);
    }
    const int w_piece2 = 16 * 9 * p.Ck * ESZ;                                     // two pieces = 16 weight rows further
    // The offsets the activation pieces of the kernel row being loaded use: recomputed when that row changes (three
    // times per tile), out-of-range for rows outside the map and -- kh >= 3 -- once there is no further group to load.
    // (Forming them at the piece, five vector instructions in front of every LDS-DMA, cost ~60 cycles of MFMA issue per
    // piece in the stamps against ~10 for a weight piece, whose offset is ready.)
    int va[NAP + 1];
    auto set_kh = [&](int kh) __attribute__((always_inline)) {
        const int tap_off = kh * p.IW * p.Cx * ESZ;
        const unsigned bad = kh >= 3 ? ~0u : (~a_flags | (kh == 0 ? a_flags >> 9 : 0u) | (kh == 2 ? a_flags >> 18 : 0u));
#pragma unroll
        for (int j = 0; j < NAP + 1; ++j) va[j] = ((bad >> j) & 1u) ? H_OOB : a_b32[j] + tap_off;
    };
    // activation piece j of (kh set by set_kh, chunk) into activation buffer `abuf`
    auto dma_a = [&](int j, int abuf, int chunk) __attribute__((always_inline)) {
        if (j == NAP && wid_u != 0) return;
        lds_t la = (lds_t)(smem + abuf * HA + (j < NAP ? wid_u * (HM / 4) + 8 * j : HM) * HROW);
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass instantiates this template too: it must not see the device builtin)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, la, 16, va[j], chunk * HROW, 0, 0);
#endif
    };
    // weight piece j of K-step (tap, chunk) into weight buffer `bbuf`
    auto dma_b = [&](int j, int bbuf, int tap, int chunk, bool live) __attribute__((always_inline)) {
        lds_t la = (lds_t)(smem + 2 * HA + bbuf * HB + (wid_u * (WN / 2) + 8 * j) * HROW);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, la, 16, live ? vb2[j & 1] : H_OOB,
                                                 (tap * p.n_chunks + chunk) * HROW + (j >> 1) * w_piece2, 0, 0);
#endif
    };

    // (S16: the MFMAs are inline assembly with the accumulator as a tied AGPR operand -- through the builtin the compiler
    // gave the 64 four-register accumulators a new destination at every MFMA and moved them back around the loop,
    // hundreds of v_accvgpr moves per K-step.  What the compiler then no longer does is count hazard wait states for
    // these MFMAs: their inputs come from ds_read (s_waitcnt, which it still places) and an accumulator is reused 63
    // MFMAs later; the epilogue reads them after a barrier.)
    typedef float accv_t __attribute__((ext_vector_type(S16 ? 4 : 16)));
    accv_t acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < (S16 ? 4 : 16); ++e) acc[i][j][e] = 0.f;

    // ---- prologue: activation tile of group 0 (kernel row 0, chunk 0), weights of K-step 0
    set_kh(0);
#pragma unroll
    for (int j = 0; j < NAP + 1; ++j) dma_a(j, 0, 0);
#pragma unroll
    for (int j = 0; j < NBP; ++j) dma_b(j, 0, 0, 0, true);

    // fragment addressing: activation row of output row (wm*WM + i*32 + lr), tap kw: + kw + 2 * segment; the second 16-row
    // block of a 32-row group (S16) is 16 rows further down: same swizzle term, a compile-time offset
    int seg_of[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) seg_of[i] = (wm * WM + i * 32) / seg;
    const int sxb = (lr >> 1) & 7;
    const int fb_base = 2 * HA + (wn * WN + lr) * HROW;

    // ---- the K loop.  K-step `it` = (kh, chunk, kw) multiplies from activation buffer (it / 3) & 1 and weight buffer
    // it & 1 in NSUB sub-steps, fragments double-buffered in registers: every group of MFMAs (one activation block
    // against all weight blocks) is followed by two fragment reads for the NEXT sub-step.  The workgroup barrier of a step
    // sits in front of the MFMAs of its LAST sub-step: by then every wave holds the step's last fragments in registers, so
    //   * after the barrier the step's weight buffer is free: the weights of step it + 2 start loading into it, and
    //   * the first fragments of step it + 1 (landed: every wave waited for its own pieces before the barrier) are read
    //     in the shadow of the last sub-step's MFMAs instead of in front of an idle MFMA pipe (in-kernel stamps,
    //     tools/row3_stamps.py: barrier -> first MFMA of the next step was ~450 of a step's ~3300 cycles).
    // A "period" (barrier to barrier) is therefore the last sub-step of one step + the others of the next = NSUB * NA
    // groups; its staging instructions are the NBP weight pieces of the step after next and -- in the period that
    // ends in the first step of a (kh, chunk) group -- the NAP (+1) activation pieces of the next group, which have
    // two more periods to land: the wait in front of the barrier leaves exactly those in flight.
    // WV >= 0 (the 256 x 256 tile): the loop is instantiated once per wave of the workgroup and wave WV issues ONE
    // staging instruction per group, a quarter of a group after the previous wave's, so that the CU's vector-memory port
    // sees one 1-KiB piece every 32 cycles instead of four at a time (stamps: issued by all four waves at the same
    // point, a piece costs each of them 50-70 cycles of MFMA issue; one wave at a time, nothing measurable).
    // WV < 0 (-DMSG_ROW3_NO_STAGGER, A/B): one loop for all waves, two pieces per group from the start of the period.
    struct Pos { int kh, chunk, kw; };
    auto k_loop = [&](auto wv_tag) __attribute__((always_inline)) {
        constexpr int WV = decltype(wv_tag)::value;
        constexpr int RPG = 3;                                      // fragment reads per group of MFMAs
        constexpr int ACT_IN_FLIGHT = NAP + (WV == 0 ? 1 : 0);      // (WV < 0: wave 0 waits for its extra piece early)
        bf16v8 fa[2][NA], fb[2][NB];
        // LDS byte offsets of the activation fragments, one per 32-row group and sub-step (buffer, row and swizzled slot
        // included; the second 16-row block of a group is a compile-time offset further): the reads carry no address
        // arithmetic
        int a_addr[MI][NSUB];
        auto frag_row = [&](int i, int kw_, int buf) __attribute__((always_inline)) {
            const int r = wm * WM + i * 32 + lr + kw_ + 2 * seg_of[i];
            const int row = buf * HA + r * HROW, sx = (r >> 1) & 7;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) a_addr[i][sub] = row + (((SLOTS * sub + lh) ^ sx) << 4);
        };
        auto read_a = [&](int sub, int ia) __attribute__((always_inline)) {
            const int i32 = ia * BLK / 32, extra = (ia * BLK % 32) * HROW;
            fa[sub & 1][ia] = *reinterpret_cast<const bf16v8*>(smem + a_addr[i32][sub] + extra);
        };
        auto read_b = [&](int sub, int jb, const char* sB) __attribute__((always_inline)) {
            fb[sub & 1][jb] = *reinterpret_cast<const bf16v8*>(sB + fb_base + jb * BLK * HROW + (((SLOTS * sub + lh) ^ sxb) << 4));
        };
        // the t-th fragment read for a sub-step, in the order its MFMAs need them: the first group takes ALL weight
        // fragments but only the first activation fragment
        auto read_nth = [&](int sub, int t, const char* sB) __attribute__((always_inline)) {
            if (t == 0) read_b(sub, 0, sB);
            else if (t == 1) read_a(sub, 0);
            else if (t <= NB) read_b(sub, t - 1, sB);
            else read_a(sub, t - NB);
        };
        auto mfma = [&](int set, int ia, int jb) __attribute__((always_inline)) {   // operands swapped: transposed accumulators (see the epilogue)
            if constexpr (S16) {
                accv_t& c = acc[ia][jb];
                const bf16v8& wf = fb[set][jb];
                const bf16v8& xf = fa[set][ia];
#if defined(__HIP_DEVICE_COMPILE__)
                asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(wf), "v"(xf));
#endif
            } else {
                acc[ia][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][jb], fa[set][ia], acc[ia][jb], 0, 0, 0);
            }
        };
        int kh = 0, chunk = 0;                                      // the (kh, chunk) group of the current K-step
        int kh_l = 0, chunk_l = 0;                                  // the next group = the one whose activation tile loads
        int abuf = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) frag_row(i, 0, 0);
#pragma unroll
        for (int t = 0; t < NA + NB; ++t) read_nth(0, t, smem);
        // the weight pieces of step 1 that a period issues behind its barrier (last sub-step's groups: one per group with
        // per-wave loops, two per group otherwise)
#pragma unroll
        for (int q = 0; q < (WV >= 0 ? NA : 2 * NA); ++q) if (q < NBP) dma_b(q, 1, 1, 0, true);

        // One K-step, the horizontal tap KW a compile-time constant: the loop below runs the three steps of a (kh, chunk)
        // group back to back, so that "this period carries the activation pieces" (KW == 0) and the count the wait leaves
        // in flight cost no branches (a skipped scalar branch per piece slot cost ~30 cycles of MFMA issue each).
        auto k_step = [&](auto kw_tag, int it) __attribute__((always_inline)) {
            constexpr int KW = decltype(kw_tag)::value;
            R3_STAMP(0);
            const int bbuf = it & 1;
            const bool live1 = it + 1 < p.n_iters, live2 = it + 2 < p.n_iters;
            const int abuf_n = KW == 2 ? abuf ^ 1 : abuf;           // activation buffer of step it + 1
            if (KW == 0) {                                          // the next group's activation tile loads in this period
                chunk_l = chunk + 1; kh_l = kh;
                if (chunk_l == p.n_chunks) { chunk_l = 0; ++kh_l; }
                if (kh_l != kh) set_kh(kh_l);       // (kh_l == 3: nothing left to load -- zeros into the idle buffer)
            }
            const Pos n1 = KW < 2 ? Pos{kh, chunk, KW + 1} : Pos{kh_l, chunk_l, 0};
            const Pos n2 = KW < 1 ? Pos{kh, chunk, 2} : Pos{kh_l, chunk_l, KW - 1};
            constexpr bool ACTS = KW == 0;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const bool tail = sub == NSUB - 1;
                if (tail) {
                    // Before the barrier: this wave's own pieces for step it + 1 have LANDED (an explicit count -- the
                    // compiler's own wait is derived from alias analysis of the DMA destinations and once left two pieces
                    // in flight: sporadically wrong output channels) and its fragment reads of this step are done.
                    if (ACTS) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(ACT_IN_FLIGHT) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    R3_STAMP(4);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    R3_STAMP(5);
                    abuf = abuf_n;                                  // from here on the fragment reads are step it + 1's
                }
                const char* sB = smem + (tail ? bbuf ^ 1 : bbuf) * HB;
#pragma unroll
                for (int ia = 0; ia < NA; ++ia) {
                    const int ps = ((sub + 1) % NSUB) * NA + ia;    // slot of the period: the tail's groups come first
                    // One MFMA, then what its gap carries (a sched_barrier per MFMA: the order written here is the order
                    // issued).  An MFMA holds the SIMD's vector issue for 8 of its 32 / 16 cycles; what else the wave issues
                    // is free only while it fits the rest, so the group's fragment reads, its staging instruction and the
                    // address arithmetic are spread one or two per gap instead of clustered behind the group.
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) {
                        mfma(sub & 1, ia, jb);
                        // fragments of the next sub-step: RPG reads per group from the first group on (the compiler's
                        // s_waitcnt in front of a sub-step's first group waits for ALL reads in flight, so the last ones
                        // must be well ahead of it), read k of the group after MFMA 2 k (8-MFMA groups) or k
#pragma unroll
                        for (int k = 0; k < RPG; ++k) {
                            const int t = RPG * ia + k;
                            if (t < NA + NB && jb == (NB == 8 ? 2 * k : (k < NB ? k : NB - 1))) read_nth((sub + 1) % NSUB, t, sB);
                        }
                        // staging: wave WV's piece of the group after MFMA 2 WV + 1 (or WV); WV < 0: two at the group's end
                        const int dj = NB == 8 ? (jb - 1) / 2 : jb;
                        const bool dslot = WV >= 0 ? (NB == 8 ? (jb & 1) && dj == WV : dj == WV) : jb == NB - 1;
                        if (dslot) {
                            const int bufw = tail ? bbuf : bbuf ^ 1;
                            const Pos q = tail ? n2 : n1;
                            const bool live = tail ? live2 : live1;
                            if (WV >= 0) {
                                if (ps < NBP) dma_b(ps, bufw, q.kh * 3 + q.kw, q.chunk, live);
                                else if (!tail && ACTS && ps - NBP < NAP) dma_a(ps - NBP, abuf ^ 1, chunk_l);
                            } else {
#pragma unroll
                                for (int t = 0; t < 2; ++t) {       // weights behind the barrier, activations from sub-step 0 on
                                    static_assert(NBP <= 2 * NA, "weight pieces fit the last sub-step's groups");
                                    if (tail) { if (2 * ps + t < NBP) dma_b(2 * ps + t, bufw, q.kh * 3 + q.kw, q.chunk, live); }
                                    else if (ACTS && 2 * (ps - NA) + t < NAP + 1) dma_a(2 * (ps - NA) + t, abuf ^ 1, chunk_l);
                                }
                            }
                        }
                        if (!tail && ACTS && WV == 0 && ps == NSUB * NA - 1 && jb == NB - 1) dma_a(NAP, abuf ^ 1, chunk_l);
                        if (sub == NSUB - 2 && jb == NB - 1) {
                            // Next step's fragment addresses, in place after the last read of this step that uses them, here
                            // (VALU in the MFMAs' shadow) rather than behind the barrier.  Activation fragment f is read
                            // t-th, t = 1 (f = 0) or NB + f, in group t / RPG; a 32-row group's address serves fragments
                            // 32 / BLK * i32 ... 32 / BLK * (i32 + 1) - 1.
#pragma unroll
                            for (int i32 = 0; i32 < MI; ++i32) {
                                const int f_last = (32 / BLK) * (i32 + 1) - 1;
                                if ((f_last == 0 ? 1 : NB + f_last) / RPG == ia) {
                                    frag_row(i32, (KW + 1) % 3, abuf_n);
#pragma unroll
                                    for (int sub2 = 0; sub2 < NSUB; ++sub2) asm volatile("" : "+v"(a_addr[i32][sub2]));
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                R3_STAMP(tail ? 6 : 1 + sub);
            }
            if (KW == 2) { kh = kh_l; chunk = chunk_l; }
        };
        R3_CLOCK(1);
        for (int it = 0; it < p.n_iters; it += 3) {                 // (n_iters = 9 * n_chunks)
            k_step(std::integral_constant<int, 0>{}, it);
            k_step(std::integral_constant<int, 1>{}, it + 1);
            k_step(std::integral_constant<int, 2>{}, it + 2);
        }
        R3_CLOCK(2);
    };
    if (STAGGER) {
        switch (wid_u) {
            case 0: k_loop(std::integral_constant<int, 0>{}); break;
            case 1: k_loop(std::integral_constant<int, 1>{}); break;
            case 2: k_loop(std::integral_constant<int, 2>{}); break;
            default: k_loop(std::integral_constant<int, 3>{}); break;
        }
    } else {
        k_loop(std::integral_constant<int, -1>{});
    }
    if (S16) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // (the last inline-assembly MFMAs retire before anything reads them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the dummy pieces of the last step write zeros: they must land first)
    __syncthreads();

    // ---- epilogue: wave-private WM x WN bf16 patch in LDS, then 16-B stores.  Transposed accumulators: lane (lr, lh)
    // owns pixel 32 i + lr and, per group g = e >> 2, the four CONSECUTIVE channels 32 j + 8 g + 4 lh + (0..3): one packed
    // 8-byte LDS write each; 8-byte unit u of row r lives at unit u ^ (r & 15).
    constexpr int PITCH = WN * ESZ;
    char* ep = smem + wid * (WM * PITCH);
    // (S16: lane (lr, lh) owns pixel 16 i + lr and the four channels 16 j + 4 lh + (0..3) of block (i, j): G = 1 group)
    constexpr int G = S16 ? 1 : 4;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int unit = S16 ? 4 * j + lh : 8 * j + 2 * g + lh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
                const int nb = n0 + wn * WN + 4 * unit;
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = bias[nb + e];              // (N % HN == 0: eligibility)
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int row = i * BLK + lr;
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
    // Eligibility (msg_conv2d_fprop_row3_eligible) leaves whole tiles only -- Mtot % HM == 0, N % HN == 0 -- and the output row of
    // pixel m is row m: no bounds predicates and no coordinates here.  (One wave per SIMD: every epilogue instruction is four
    // cycles that nothing overlaps; the stepped (b, oh, ow) coordinates and 16 predicated stores per half were a third of them.)
    // half-patches of 64 rows: all rows / residual vectors of a half requested before any is used
    if constexpr (EPI == 1) {
        // ---- activation backward in the epilogue (ActEpilogue::enabled == 3).  Per half patch: sign source (bytes or the stored
        // output), optional residual, y = (conv + residual) * slope in fp32 from the ROUNDED conv result, one rounding; the lane's
        // eight channel sums and its noise-weighted sum run across the halves.
        float cs[VEC], sn = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) cs[e] = 0.f;
        const unsigned char* ep_sign = reinterpret_cast<const unsigned char*>(smem + SIGN_OFF) + (wn * WN + ec) / 8;
#pragma unroll 1
        for (int half = 0; half < MI / 2; ++half) {
            const int m_first = m0 + wm * WM + half * 64 + er;
            const long long qg = (p.per_sample ? (long long)bz * ohw : 0) + m_first;     // pixel index over the whole batch
            float a_noise[NP];
            if (p.act.noise) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) a_noise[pass] = ep_noise[wm * WM + half * 64 + pass * RPP + er];
            }
            if (half == 0) __syncthreads();
            u32x4 v[NP];
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int row = half * 64 + pass * RPP + er;
                v[pass] = *reinterpret_cast<const u32x4*>(ep + row * PITCH + ((u16 ^ ((row & 15) >> 1)) << 4));
            }
            unsigned sbits[NP];
            if (p.act.mask) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) sbits[pass] = ep_sign[(wm * WM + half * 64 + pass * RPP + er) * (HN / 8)];
            } else {
                const bf16_t* srow = reinterpret_cast<const bf16_t*>(p.act.sign_src) + n + qg * p.act.sign_ld;
                u32x4 sv[NP];
#pragma unroll
                for (int pass = 0; pass < NP; ++pass)
                    sv[pass] = *reinterpret_cast<const u32x4*>(srow + (long long)(pass * RPP) * p.act.sign_ld);
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) sbits[pass] = act_sign_byte(sv[pass]);
            }
            u32x4 r[NP];
            const bool has_res = p.act.residual != nullptr;
            if (has_res) {
                const bf16_t* rrow = reinterpret_cast<const bf16_t*>(p.act.residual) + n + qg * p.act.res_ld;
#pragma unroll
                for (int pass = 0; pass < NP; ++pass)
                    r[pass] = *reinterpret_cast<const u32x4*>(rrow + (long long)(pass * RPP) * p.act.res_ld);
            }
            if (er & 1) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) v[pass] = u32x4{v[pass][2], v[pass][3], v[pass][0], v[pass][1]};
            }
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
#pragma clang fp contract(off)
                float f[VEC], rowsum = 0.f;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const unsigned w = v[pass][e >> 1];
                    float val = (e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
                    if (has_res) {
                        const unsigned wr = r[pass][e >> 1];
                        val += (e & 1) ? __uint_as_float(wr & 0xffff0000u) : __uint_as_float(wr << 16);
                        val = bf2f(f2bf(val));         // (the two-pass form stores the sum before the activation backward reads it)
                    }
                    f[e] = val * p.act.scale * (((sbits[pass] >> e) & 1u) ? 1.f : p.act.alpha);   // (the stand-alone kernel's order: same bits)
                    cs[e] += f[e];
                    rowsum += f[e];
                }
                if (p.act.noise) sn = fmaf(rowsum, a_noise[pass], sn);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[pass][e] = (unsigned)f2bf(f[2 * e]) | ((unsigned)f2bf(f[2 * e + 1]) << 16);
            }
            bf16_t* yrow = y + (p.per_sample ? (long long)bz * p.y_bstride : 0) + n + (long long)m_first * p.ldy;
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) *reinterpret_cast<u32x4*>(yrow + (long long)(pass * RPP) * p.ldy) = v[pass];
        }
        // partial sums: the lanes that share a channel vector (lane = er * LPR + u16) in a fixed butterfly; one row of part_b per
        // (sample, pixel tile, wave row), one entry of part_n per (sample, tile, wave) -- summed in index order by the reduce launch
        const long long mt = (long long)(p.per_sample ? bz : 0) * p.m_tiles + L / p.n_tiles;
        if (p.act.part_b) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float t = cs[e];
#pragma unroll
                for (int off = 32; off >= LPR; off >>= 1) t += __shfl_xor(t, off, 64);
                cs[e] = t;
            }
            if (er == 0) {
                float* dst = p.act.part_b + (mt * 2 + wm) * p.N + n;
#pragma unroll
                for (int e = 0; e < VEC; ++e) dst[e] = cs[e];
            }
        }
        if (p.act.part_n) {
            sn = wave_sum(sn);
            if (lane == 0) p.act.part_n[(mt * p.n_tiles + L % p.n_tiles) * 4 + wid] = sn;
        }
        R3_CLOCK(3);
        return;
    }
#pragma unroll 1
    for (int half = 0; half < MI / 2; ++half) {
        const int m_first = m0 + wm * WM + half * 64 + er;              // this lane's first pixel of the half (inside the sample)
        float a_bias[VEC], a_noise[NP];
        if (p.act.enabled == 1) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) a_bias[e] = ep_bias[wn * WN + ec + e];
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) a_noise[pass] = ep_noise[wm * WM + half * 64 + pass * RPP + er];
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
            if (p.act.alpha >= 0.f && p.act.alpha <= 1.f) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass)
                    v[pass] = act_epilogue_apply<bf16_t, true>(v[pass], a_bias, a_noise[pass], p.act.alpha, p.act.scale);
            } else {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass)
                    v[pass] = act_epilogue_apply<bf16_t>(v[pass], a_bias, a_noise[pass], p.act.alpha, p.act.scale);
            }
        } else if (p.act.enabled == 2) {
            const bf16_t* rrow = reinterpret_cast<const bf16_t*>(p.act.residual) + n +
                                 ((p.per_sample ? (long long)bz * ohw : 0) + m_first) * p.act.res_ld;
            u32x4 r[NP];
#pragma unroll
            for (int pass = 0; pass < NP; ++pass)
                r[pass] = *reinterpret_cast<const u32x4*>(rrow + (long long)(pass * RPP) * p.act.res_ld);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) v[pass] = residual_epilogue_apply<bf16_t>(v[pass], r[pass], p.act.res_gain);
        }
        bf16_t* yrow = y + (p.per_sample ? (long long)bz * p.y_bstride : 0) + n + (long long)m_first * p.ldy;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) *reinterpret_cast<u32x4*>(yrow + (long long)(pass * RPP) * p.ldy) = v[pass];
        if (p.act.enabled == 1 && p.act.mask) {
            // sign bytes for the activation's backward, in this workgroup's own block of the map (act_mask_index with
            // tile_m = HM, tile_n = HN)
            const long long tile = (long long)((p.per_sample ? (long long)bz * ohw : 0) + m0) / HM * p.n_tiles + n0 / HN;
            unsigned char* mb = p.act.mask + tile * (HM * (HN / 8)) + (wn * WN + ec) / 8 + (wm * WM + half * 64 + er) * (HN / 8);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) mb[pass * RPP * (HN / 8)] = (unsigned char)act_sign_byte(v[pass]);
        }
    }
    R3_CLOCK(3);
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
    static const int enabled = msg_tunable("MSG_CONV_ROW3", 1);
    if (!enabled || kh != 3 || kw != 3 || IH != OH || IW != OW || Ck % 64 || Cx % 64) return 0;
    const bool per_sample = w_batch_stride != 0;
    const long long mtot = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    // The 128 x 128 variant (two workgroups per CU) takes the layers with 128 / 384 output channels from the plain 128x128
    // kernel: 3x3 128->128 @256^2 516 -> 476 us, 256->128 @256^2 866 -> 819, 256->384 @128^2 650 -> 614, 384->384 @64^2
    // 235 -> 205.  MSG_CONV_ROW3_NARROW=0 switches it off (A/B).
    static const int narrow = msg_tunable("MSG_CONV_ROW3_NARROW", 1);
    // 32-wide maps: the 128 x 128 tile holds four image rows (4 x 34 = 136 buffer rows) whatever the channel count:
    // 3x3 768->768 @32^2, B=16: 229 -> 179 us, 1024->768: 318 -> 272 (they ran on the ping-pong kernel).  MSG_CONV_ROW3_W32=0: off.
    static const int w32 = msg_tunable("MSG_CONV_ROW3_W32", 1);
    int hn = row3_tile_columns(N);
    // short K (few channels): a 256 x 256 tile's prologue and epilogue are not amortised and nothing overlaps them with one
    // workgroup per CU; the 128 x 128 tile runs two.  MSG_CONV_ROW3_SHORTK = largest Ck that prefers the small tile.
    // (round 3: 3x3 128->256 @256^2, B=32: 632 -> 609 us on the small tile.  Round 4, after the large tile's prologue and epilogue
    //  lost a third of their instructions, the same launch in isolation: B=32 1 166 vs 1 077 us, B=16 630 vs 594 us in favour of the
    //  LARGE tile; in the training step the two launches per iteration it concerns are 0.13 ms -- below what two runs on one box
    //  differ by -- and were left where they are)
    static const int shortk = msg_tunable("MSG_CONV_ROW3_SHORTK", 128);
    if (hn == 256 && Ck <= shortk && N % 128 == 0) hn = 128;
    if (OW == 32 && w32 && N >= 128 && (long long)((N + 127) / 128) * 128 * 100 <= (long long)N * 115) hn = 128;
    if (!hn || (hn == 128 && !narrow) || N % hn) return 0;
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
    p.seg_magic = 65536 / (p.seg_len + 2) + 1;
    for (int row = 0; row < hm + 16; ++row)
        if (((row * p.seg_magic) >> 16) != row / (p.seg_len + 2)) return 0;      // (cannot happen for rows < 2^8 * 2; the plain kernels take it)
    auto log2_exact = [](long long v) { int sh = 0; while ((1ll << sh) < v) ++sh; return (1ll << sh) == v ? sh : -1; };
    p.ow_shift = log2_exact(OW);
    p.ohw_shift = log2_exact((long long)OH * OW);
    if (p.ohw_shift < 0) p.ow_shift = -1;
    p.n_tiles = (N + hn - 1) / hn;
    const long long blocks = (long long)p.m_tiles * p.n_tiles;
    dim3 grid((unsigned)blocks, 1, per_sample ? B : 1);
    // MSG_CONV_ROW3_S16=0: the 256 x 256 tile on v_mfma_f32_32x32x16_bf16 (A/B)
    static const int s16 = msg_tunable("MSG_CONV_ROW3_S16", 1);
    static const int s16n = msg_tunable("MSG_CONV_ROW3N_S16", 1);    // (the 128 x 128 tile: +2..6 %)
    if (p.act.enabled == 3) {
        if (bias) return 0;
        if (hn == 256)
            hipLaunchKernelGGL((conv_fprop_row3_kernel<4, 4, true, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)w, (bf16_t*)y, bias, p);
        else
            hipLaunchKernelGGL((conv_fprop_row3_kernel<2, 2, true, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)w, (bf16_t*)y, bias, p);
        return 1;
    }
    if (hn == 256 && s16)
        hipLaunchKernelGGL((conv_fprop_row3_kernel<4, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    else if (hn == 256)
        hipLaunchKernelGGL((conv_fprop_row3_kernel<4, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    else if (s16n)
        hipLaunchKernelGGL((conv_fprop_row3_kernel<2, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    else
        hipLaunchKernelGGL((conv_fprop_row3_kernel<2, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    return 1;
}
