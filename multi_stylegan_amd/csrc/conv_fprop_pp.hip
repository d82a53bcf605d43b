// a3/a4: large-tile "ping-pong" variant of the implicit-GEMM convolution (same contract as conv_fprop.hip, bf16 only).
//
// The 128x128 kernel keeps the matrix pipe ~50 % busy: per MFMA it needs 1 KiB of LDS fragment reads and 0.5 KiB of
// LDS fills, and its four waves sit in the same barrier at the same time.  This kernel changes the ratios and the
// schedule:
//   * workgroup tile 256 (pixels) x 256 (channels) x 64 (K), 8 waves, wave tile 128 x 64 -> 0.5 KiB of fragment reads
//     and 0.25 KiB of LDS-DMA fill per MFMA; 2 LDS stages of 64 KiB; one workgroup per CU;
//   * the two waves that share a SIMD (w and w+4) run half a phase apart: while one issues its 16 MFMAs (a 64x64
//     half of its tile) the other reads the 16 fragments of its next half and issues LDS-DMA for the next K-tile, then
//     they swap.  Four s_barrier slots per K-tile keep the two groups in lock-step; the barriers are raw (no implicit
//     vmcnt(0)), each wave waits for its own DMA pieces exactly once per K-tile, one slot before they are needed.
//
// Measured with in-kernel s_memtime stamps (tools/pp_stamps.py; round 4, steady state = the last workgroups of a launch,
// profiles/r04_pp_stamps.txt, 2x2 stride-2 512 -> 512 @256^2): MFMA phase ~690 cycles, fragment-read phase ~390-470, and issuing
// a wave's 8 LDS-DMA pieces ~630-680 (~80 per piece with four waves issuing together; ~850 before the pieces were addressed
// through a buffer descriptor instead of 64 per-lane 64-bit addresses).  Phase A (issue + reads, ~1050) against the other
// group's MFMA phase (~690) is the critical path: a K-tile takes ~3500 cycles (2048 of MFMA), ~4500 when the tap changes
// (the new tap's rows come from beyond L2 and two stages leave them one K-tile to land); a workgroup of 32 K-tiles spends
// 3.8 us before, 67 us in and 5.6 us behind its K loop, at a 1.9 GHz clock.  Spreading the pieces into the MFMA phases:
// a first attempt ran out of VGPRs (128 accumulators + 64 fragment registers leave ~60 for everything else); four of the
// eight behind every fourth MFMA of phase B (mfma_half's `dma` argument) measured no different in round 4; nor did a row
// assignment that gives every wave 16 rows of the half read first and 16 of the half read three slots later, with the
// latter's two pieces moved to phase C behind counted waits (six urgent + two deferred pieces per wave: bit-correct, same time).
//
//   slot (global)      4t        4t+1      4t+2      4t+3      4t+4
//   group 0 (w<4)    read h0(t)  MFMA h0   read h1   MFMA h1   read h0(t+1) ...
//   group 1 (w>=4)   MFMA h1(t-1) read h0(t) MFMA h0  read h1   MFMA h1(t)  ...
//   DMA(t+1) is issued at the top of each wave's "read h0(t)" phase and retired before the barrier that ends slot
//   4t+3; the stage it fills was last read in slot 4t-1, and read phases drain lgkmcnt before their barrier.
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* lds_t;

struct ConvParamsPP {
    int B, IH, IW, Cx, Ck, OH, OW, N, ldy;
    int kh, kw, stride, pad, in_up, pixel_shuffle, per_sample;
    long long x_bstride, w_bstride, y_bstride;
    int Mtot, n_chunks, n_iters, m_tiles, n_tiles;
    ActEpilogue act;
};

constexpr int PM = 256, PN = 256, PROW = 128;
constexpr int PSTAGE = (PM + PN) * PROW;                  // 64 KiB

__device__ __forceinline__ int pswz(int row, int slot) { return row * PROW + ((slot ^ ((row >> 1) & 7)) << 4); }
#ifdef MSG_PP_STAMPS
// diagnostic build only: cycle stamps of K-tile 8 of every wave of the LAST 256 workgroups of the launch (tools/pp_stamps.py)
__device__ unsigned long long g_pp_stamps[256 * 8 * 10];
#define PP_STAMP(k) do { if (t == 8 && blockIdx.x + 256 >= gridDim.x && blockIdx.z == gridDim.z - 1 && lane == 0) { unsigned long long tt; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory"); \
    g_pp_stamps[((blockIdx.x + 256 - gridDim.x) * 8 + wid_u) * 10 + (k)] = tt; } } while (0)
extern "C" int msg_pp_debug_read(void* host_dst, int nbytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pp_stamps), nbytes) == hipSuccess ? 0 : -1;
}
// shader cycles (s_memtime) and the 100 MHz reference counter (s_memrealtime) at kernel entry (0), at the start (1) and the end
// (2) of the K loop and at kernel exit (3), every wave of the LAST 256 workgroups of the launch
__device__ unsigned long long g_pp_clock[256 * 8 * 8];
#define PP_CLOCK(k) do { if (blockIdx.x + 256 >= gridDim.x && blockIdx.z == gridDim.z - 1 && lane == 0) { unsigned long long tc, tr; \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc), "=s"(tr) :: "memory"); \
    const unsigned cl_ = blockIdx.x + 256 - gridDim.x; \
    g_pp_clock[(cl_ * 8 + wid_u) * 8 + 2 * (k)] = tc; g_pp_clock[(cl_ * 8 + wid_u) * 8 + 2 * (k) + 1] = tr; } } while (0)
// the start of K-tiles 8..23 (their period), same workgroups
__device__ unsigned long long g_pp_period[256 * 8 * 16];
#define PP_PERIOD() do { if (t >= 8 && t < 24 && blockIdx.x + 256 >= gridDim.x && blockIdx.z == gridDim.z - 1 && lane == 0) { unsigned long long tt; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory"); \
    g_pp_period[((blockIdx.x + 256 - gridDim.x) * 8 + wid_u) * 16 + (t - 8)] = tt; } } while (0)
extern "C" int msg_pp_period_read(void* host_dst, int nbytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pp_period), nbytes) == hipSuccess ? 0 : -1;
}
extern "C" int msg_pp_clock_read(void* host_dst, int nbytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pp_clock), nbytes) == hipSuccess ? 0 : -1;
}
#else
#define PP_STAMP(k) do {} while (0)
#define PP_CLOCK(k) do {} while (0)
#define PP_PERIOD() do {} while (0)
#endif
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(512, 2) void conv_fprop_pp_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ y, const float* __restrict__ bias,
                                                               ConvParamsPP p) {
    constexpr int VEC = 8, BKE = 64, ESZ = 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * PSTAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int grp = wid_u >> 2;                           // 0: waves 0-3 (tile rows 0..127), 1: waves 4-7 (rows 128..255)
    const int wn = wid_u & 3;                             // 64-column slice of the tile
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (int)(L % p.n_tiles) * PN;
    const int m0 = (int)(L / p.n_tiles) * PM;
    const int bz = blockIdx.z;
    PP_CLOCK(0);

    // ---- LDS-DMA assignment: wave w fills rows 32 w + 8 j + (lane >> 3) of A and of B (j = 0..3).  lane & 7 is the
    // PHYSICAL 16-B slot; the lane fetches the logical slot that the XOR swizzle puts there.  Row coordinates are
    // recomputed at every tap change (once per n_chunks K-tiles) instead of living in registers: this kernel needs its
    // VGPRs for 128 accumulators + 64 fragment registers.
    const int slot_phys = lane & 7;
    const int row0 = wid * 32 + (lane >> 3);                 // row of j = 0; j adds 8
    const int sw0 = (row0 >> 1) & 7;                         // swizzle of rows j = 0, 2 ; rows j = 1, 3 use sw0 ^ 4
    const int ohw = p.OH * p.OW;
    const int taps = p.kh * p.kw;
    // (32-bit buffer offsets + two descriptors instead of 64-bit pointers per row: a 1-KiB LDS-DMA piece addressed through a
    //  descriptor costs the texture-address unit about half of what a piece with 64 per-lane 64-bit addresses does -- stamps:
    //  ~100 cycles per piece with four waves issuing together before, and the eight pieces at the top of phase A were this
    //  kernel's critical path.  Rows that must read zeros take an out-of-range offset; eligibility keeps the tensors below 2 GiB.)
    const msg_desc_t d_x = msg_make_desc((const char*)x + (p.per_sample ? (long long)bz * p.x_bstride * ESZ : 0));
    const msg_desc_t d_w = msg_make_desc((const char*)w + (p.per_sample ? (long long)bz * p.w_bstride * ESZ : 0));
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_t)smem;
    // (b, oh, ow) of this lane's four A rows, packed 10|11|11 bits (one VGPR per row); bit 31 = row is beyond M
    unsigned rc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + row0 + 8 * j;
        const bool ok = m < p.Mtot;
        const int mm = ok ? m : 0;
        const int b = p.per_sample ? 0 : mm / ohw;
        const int pix = p.per_sample ? mm : mm - b * ohw;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        rc[j] = (ok ? 0u : 0x80000000u) | ((unsigned)b << 22) | ((unsigned)oh << 11) | (unsigned)ow;
    }
    int va[4], vb[4];                                     // byte offsets of chunk 0 of the current tap (A) / of K-tile 0 (B)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + row0 + 8 * j;
        const int sl = slot_phys ^ (sw0 ^ ((j & 1) << 2));
        vb[j] = n < p.N ? (int)(((long long)n * taps * p.Ck + sl * VEC) * ESZ) : MSG_DMA_OOB;
        va[j] = MSG_DMA_OOB;
    }
    const bool ragged = (p.Cx % BKE) != 0;
    int ld_tap = -1, ld_chunk = p.n_chunks - 1, ld_tile = -1;
    auto advance = [&]() __attribute__((always_inline)) {     // cursor to the next K-tile; new tap -> new row offsets
        if (++ld_tile >= p.n_iters) {                         // past the last K-tile: zeros into an unused stage
#pragma unroll
            for (int j = 0; j < 4; ++j) { va[j] = MSG_DMA_OOB; vb[j] = MSG_DMA_OOB; }
            ld_chunk = 0;
            ld_tap = taps;                                    // (never a real tap again)
            return;
        }
        if (++ld_chunk == p.n_chunks) {
            ld_chunk = 0;
            ++ld_tap;
            const int kh_ = ld_tap / p.kw, kw_ = ld_tap - kh_ * p.kw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sl = slot_phys ^ (sw0 ^ ((j & 1) << 2));
                bool ok = (rc[j] >> 31) == 0;
                const int b = p.per_sample ? 0 : (int)((rc[j] >> 22) & 0x1ff);
                const int oh = (int)((rc[j] >> 11) & 0x7ff), ow = (int)(rc[j] & 0x7ff);
                int ih = oh * p.stride - p.pad + kh_, iw = ow * p.stride - p.pad + kw_;
                ok = ok & (ih >= 0) & (iw >= 0);
                if (p.in_up > 1) {
                    ok = ok & (ih % p.in_up == 0) & (iw % p.in_up == 0);
                    ih /= p.in_up; iw /= p.in_up;
                }
                ok = ok & (ih < p.IH) & (iw < p.IW);
                va[j] = ok ? (int)(((long long)b * p.x_bstride + ((long long)ih * p.IW + iw) * p.Cx + sl * VEC) * ESZ) : MSG_DMA_OOB;
            }
        }
    };
    // one A piece / one B piece of row group j of the K-tile the cursor stands on
    auto dma_a = [&](int j, int stage) __attribute__((always_inline)) {
        const bool c_bad = ragged && (ld_chunk * BKE + (slot_phys ^ (sw0 ^ ((j & 1) << 2))) * VEC + VEC > p.Cx);
        msg_dma16(d_x, lds0 + stage * PSTAGE + (wid_u * 32 + 8 * j) * PROW, c_bad ? MSG_DMA_OOB : va[j], ld_chunk * PROW);
    };
    auto dma_b = [&](int j, int stage) __attribute__((always_inline)) {
        msg_dma16(d_w, lds0 + stage * PSTAGE + PM * PROW + (wid_u * 32 + 8 * j) * PROW, vb[j], min(ld_tile, p.n_iters) * PROW);
    };
    auto issue = [&](int jlo, int stage) __attribute__((always_inline)) {   // two of this wave's four row groups
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) { dma_a(jlo + jj, stage); dma_b(jlo + jj, stage); }
    };
    auto issue_one = [&](int j, bool b_rows, int stage) __attribute__((always_inline)) {
        if (!b_rows) dma_a(j, stage); else dma_b(j, stage);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    bf16v8 fa[2][4], fb[2][4];                              // fragments of one 64x64 half: [tile][k-step]
    // Fragment addresses: every row this lane reads is (multiple of 32) + lr, so the swizzle term is the same for all
    // of them and the address is  [per-k-step VGPR] + compile-time offset(half, tile).  The four VGPRs per operand
    // flip between the two LDS stages with one XOR per K-tile; the reads themselves cost no address arithmetic.
    unsigned fa_addr[4], fb_addr[4];
    {
        const unsigned sw = (unsigned)(lr >> 1) & 7u;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned koff = (((unsigned)(2 * ks + lh)) ^ sw) << 4;
            fa_addr[ks] = (unsigned)((grp * 128 + lr) * PROW) + koff;
            fb_addr[ks] = (unsigned)(PM * PROW + (wn * 64 + lr) * PROW) + koff;
        }
    }
    auto read_half = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t][ks] = *reinterpret_cast<const bf16v8*>(smem + fa_addr[ks] + (h * 64 + t * 32) * PROW);
                fb[t][ks] = *reinterpret_cast<const bf16v8*>(smem + fb_addr[ks] + (t * 32) * PROW);
            }
    };
    auto flip_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa_addr[ks] ^= (unsigned)PSTAGE; fb_addr[ks] ^= (unsigned)PSTAGE; }
    };
    // 16 MFMAs of one 64x64 half.  With `dma` (wave-uniform) the four pieces of row groups 2 and 3 are issued in
    // their shadow, one after every fourth MFMA: an LDS-DMA issue costs ~100 cycles in a read phase but hides
    // between MFMAs.
    auto mfma_half = [&](int h, bool dma, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[h * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][ks], fa[i][ks], acc[h * 2 + i][j], 0, 0, 0);   // D^T: see epilogue
            if (dma) issue_one(2 + (ks >> 1), (ks & 1) != 0, stage);
        }
    };

    const int T = p.n_iters;
    // ---- prologue: K-tile 0 into stage 0 (all waves)
    advance();
    issue(0, 0);
    issue(2, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();                               // stagger: group 1 runs one slot behind group 0

    // DMA(t+1) goes to the stage that held K-tile t-1, last read in global slot 4t-1.  Every read phase ends with
    // lgkmcnt(0) BEFORE its barrier (the reading wave would idle there anyway), so once a barrier is passed all
    // fragment reads of earlier slots have returned and the stage may be refilled.  Each wave issues its 8 DMA pieces
    // at the top of its phase A (global slot 4t for group 0, 4t+1 for group 1) -- MFMA phases carry nothing but
    // MFMAs -- and retires them before the barrier that ends global slot 4t+3.
    // Measured alternatives (tools/pp_stamps.py, same shape): splitting the pieces 4 + 4 over phases A and C (group 0)
    // / A and B-between-MFMAs (group 1) was slower (994 vs 1073 TFLOP/s): an issue costs ~40 cycles between MFMAs but
    // ~150-200 in a read phase that also carries fragment reads, and only group 1 has two MFMA phases inside the window
    // in which the target stage is free.  Kept: all 8 pieces at the top of phase A.
    PP_CLOCK(1);
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        // ---- phase A: DMA for the next K-tile, fragments of half 0
        PP_PERIOD();
        PP_STAMP(0);
        if (t + 1 < T) { advance(); issue(0, buf ^ 1); issue(2, buf ^ 1); }
        PP_STAMP(1);
        read_half(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_STAMP(2);
        PP_BARRIER();
        PP_STAMP(3);
        // ---- phase B: MFMAs of half 0
        __builtin_amdgcn_s_setprio(1);
        mfma_half(0, false, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(4);
        PP_BARRIER();
        PP_STAMP(5);
        // ---- phase C: fragments of half 1
        read_half(1);
        flip_stage();                                      // next reads come from the other stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_STAMP(6);
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of K-tile t+1 is in LDS
        PP_STAMP(7);
        PP_BARRIER();
        PP_STAMP(8);
        // ---- phase D: MFMAs of half 1
        __builtin_amdgcn_s_setprio(1);
        mfma_half(1, false, 0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_STAMP(9);
        PP_BARRIER();
    }
    if (grp == 0) PP_BARRIER();
    PP_CLOCK(2);

    // ---- epilogue: wave-private 128 x 64 bf16 patch in LDS (16 KiB per wave = all 128 KiB), then 16-B stores.
    // The MFMAs were issued with the operands swapped, so an accumulator block holds the TRANSPOSED product: lane
    // (lr, lh) owns pixel 32 i + lr and, per group g = e >> 2, the four CONSECUTIVE channels 32 j + 8 g + 4 lh + (0..3).
    // They go to LDS as one packed 8-byte write (32 ds_write_b64 per wave instead of 128 ds_write_b16); 8-byte unit u
    // of row r lives at unit u ^ (r & 15), which spreads the 16 rows of a lane group over all banks.
    constexpr int PITCH = 64 * ESZ;
    char* ep = smem + wid * (128 * PITCH);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int unit = 8 * j + 2 * g + lh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
                const int nb = n0 + wn * 64 + 4 * unit;
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = nb + e < p.N ? bias[nb + e] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 32 + lr;
                uint2 pk;
                pk.x = (unsigned)f2bf(acc[i][j][4 * g + 0] + bv[0]) | ((unsigned)f2bf(acc[i][j][4 * g + 1] + bv[1]) << 16);
                pk.y = (unsigned)f2bf(acc[i][j][4 * g + 2] + bv[2]) | ((unsigned)f2bf(acc[i][j][4 * g + 3] + bv[3]) << 16);
                *reinterpret_cast<uint2*>(ep + row * PITCH + ((unit ^ (row & 15)) << 3)) = pk;
            }
        }
    const int er = lane >> 3, ec = (lane & 7) * VEC;          // 8 lanes per 64-channel row, 8 rows per pass
    const int n = n0 + wn * 64 + ec;
    int nn = n, q = 0;
    if (p.pixel_shuffle) { const int oc = p.N >> 2; q = n / oc; nn = n - q * oc; }
    const int lim = p.pixel_shuffle ? (p.N >> 2) - nn : p.N - n;  // valid elements left in this channel run
    const bool n_ok = n < p.N;
    // Row coordinates (sample, oh, ow) of this lane's row of pass 0 by division ONCE; every later pass is 8 pixels
    // further along and is reached by stepping.  gp[pass] = index of the OUTPUT pixel the row is stored to (sample-
    // major, pixel-shuffled where asked), -1 for rows past M; the fused stage's bias vector and per-row noise values
    // are fetched in the same sweep, before the barrier.
    int gp[16];
    float a_bias[VEC], a_noise[16];
    {
        const int m_first = min(m0 + grp * 128 + er, p.Mtot - 1);
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
        for (int pass = 0; pass < 16; ++pass) {
            const bool ok = m0 + grp * 128 + pass * 8 + er < p.Mtot;
            const int pix = oh * p.OW + ow;
            const int g = p.pixel_shuffle ? ((b * 2 * p.OH + 2 * oh + (q >> 1)) * (2 * p.OW) + 2 * ow + (q & 1))
                                          : b * ohw + pix;
            gp[pass] = ok ? g : -1;
            a_noise[pass] = (want_noise && ok) ? nw * p.act.noise[(long long)(p.act.noise_batch == 1 ? 0 : b) * ohw + pix] : 0.f;
            ow += 8;
            while (ow >= p.OW) { ow -= p.OW; ++oh; }
            if (!p.per_sample) while (oh >= p.OH) { oh -= p.OH; ++b; }
        }
    }
    __syncthreads();
    // All 16 patch rows are read (and, for the residual merge, all 16 residual vectors requested) BEFORE any of them is
    // used: one LDS / HBM latency per tile instead of one per pass.  16-byte unit (lane & 7) of a row = 8-byte units
    // 2 (lane & 7), +1, stored at (unit ^ (row & 15)): one aligned 16-byte read, halves swapped when the row is odd.
    u32x4 v[16];
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
        const int row = pass * 8 + er;
        v[pass] = *reinterpret_cast<const u32x4*>(ep + row * PITCH + (((lane & 7) ^ ((row & 15) >> 1)) << 4));
    }
    if (er & 1) {
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) v[pass] = u32x4{v[pass][2], v[pass][3], v[pass][0], v[pass][1]};
    }
    if (p.act.enabled == 1) {
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) v[pass] = act_epilogue_apply<bf16_t>(v[pass], a_bias, a_noise[pass], p.act.alpha, p.act.scale);
    } else if (p.act.enabled == 2) {                          // residual merge (never with pixel_shuffle)
        const bf16_t* rbase = reinterpret_cast<const bf16_t*>(p.act.residual) + (n_ok ? n : 0);
        u32x4 r[16];
#pragma unroll
        for (int pass = 0; pass < 16; ++pass)
            r[pass] = *reinterpret_cast<const u32x4*>(rbase + (long long)max(gp[pass], 0) * p.act.res_ld);
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) v[pass] = residual_epilogue_apply<bf16_t>(v[pass], r[pass], p.act.res_gain);
    }
    bf16_t* ybase = y + (p.pixel_shuffle ? nn : n);
    if (n_ok && lim >= VEC) {
#pragma unroll
        for (int pass = 0; pass < 16; ++pass)
            if (gp[pass] >= 0) *reinterpret_cast<u32x4*>(ybase + (long long)gp[pass] * p.ldy) = v[pass];
    } else if (n_ok) {                                        // ragged channel tail: element stores
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            if (gp[pass] < 0) continue;
            bf16_t* dst = ybase + (long long)gp[pass] * p.ldy;
            for (int e = 0; e < lim; ++e) dst[e] = (bf16_t)(v[pass][e >> 1] >> (16 * (e & 1)));
        }
    }
    PP_CLOCK(3);
}

// Which shapes take the large tile (shared by the launcher below and by msg_conv2d_fprop_plan).
extern "C" int msg_conv2d_fprop_pp_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                                            int kh, int kw, long long w_batch_stride) {
    static const int enabled = msg_tunable("MSG_CONV_PP", 1);
    if (!enabled) return 0;
    const bool per_sample = w_batch_stride != 0;
    const long long mtot = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    const int n_iters = kh * kw * (Ck / 64);
    if (N < 256 || mtot < 1024 || n_iters < 4 || mtot >= (1ll << 31)) return 0;
    // output-channel counts that leave a mostly empty last 256-column tile (N = 384: 1.5 tiles, a third of the MFMA
    // work on padding) run faster on the 128-wide tile: 256->384 @128^2 532 vs 660 us
    if ((long long)((N + PN - 1) / PN) * PN * 100 > (long long)N * 115) return 0;
    if ((long long)(n_iters + 1) * PROW + 128 > 65536) return 0;
    // 31-bit buffer offsets: the activations of one launch (of one sample with per-sample weights) and one weight set
    if ((long long)(per_sample ? 1 : B) * IH * IW * Cx * 2 >= 0x7ffffff0ll || (long long)N * kh * kw * Ck * 2 >= 0x7ffffff0ll) return 0;
    const long long blocks = ((mtot + PM - 1) / PM) * ((N + PN - 1) / PN);
    if (blocks >= (1ll << 31)) return 0;
    if (blocks * (per_sample ? B : 1) < 224) return 0;      // one workgroup per CU: small grids belong to the 128-tile kernel
    return 1;
}

// Called by msg_conv2d_fprop (conv_fprop.hip) for shapes where the large tile pays; returns 1 if it launched.
extern "C" int msg_conv2d_fprop_pp_try(const void* x, const void* w, const float* bias, void* y,
                                       int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                       int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                       long long w_batch_stride, const ActEpilogue* act, void* stream) {
    if (!msg_conv2d_fprop_pp_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride)) return 0;
    const bool per_sample = w_batch_stride != 0;
    const long long mtot = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    const int n_iters = kh * kw * (Ck / 64);
    ConvParamsPP p{};
    p.B = B; p.IH = IH; p.IW = IW; p.Cx = Cx; p.Ck = Ck; p.OH = OH; p.OW = OW; p.N = N; p.ldy = ldy;
    p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.in_up = in_up; p.pixel_shuffle = pixel_shuffle;
    p.per_sample = per_sample;
    if (act) p.act = *act;
    p.x_bstride = (long long)IH * IW * Cx;
    p.w_bstride = w_batch_stride;
    p.y_bstride = pixel_shuffle ? 4ll * OH * OW * ldy : (long long)OH * OW * ldy;
    p.Mtot = (int)mtot;
    p.n_chunks = Ck / 64;
    p.n_iters = n_iters;
    p.m_tiles = (int)((mtot + PM - 1) / PM);
    p.n_tiles = (N + PN - 1) / PN;
    const long long blocks = (long long)p.m_tiles * p.n_tiles;
    dim3 grid((unsigned)blocks, 1, per_sample ? B : 1);
    hipLaunchKernelGGL(conv_fprop_pp_kernel, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, bias, p);
    return 1;
}
