// a3/a4: implicit-GEMM convolution on the gfx950 matrix cores, channels-last.
//
//   Y[b, oh, ow, n] = bias[n] + sum_{kh,kw,c} X[b, (oh*stride + kh - pad)/in_up, (ow*stride + kw - pad)/in_up, c]
//                                            * W[(b)][n][kh][kw][c]
//
// GEMM view: M = output pixels, N = output channels, K = (tap, channel).  One kernel serves every dense contraction
// of the path: 3x3/1x1 "same" convs, the strided 3x3 of the discriminator, the 2x2 stride-2 transposed conv of the
// generator (as a 1x1 conv to 4*O channels written pixel-shuffled), and -- with the weights re-laid by the caller --
// all of their data gradients (in_up = 2 expresses a transposed strided conv as a gather with parity holes).
// Weights may be shared by the batch (batch folded into M) or per sample (grid.z = sample): the latter is how the
// modulated/demodulated convolution runs (multi_stylegan_generator.py:384-411 builds one weight set per sample).
//
// Tiling (wave64, MFMA 32x32): workgroup 128(M) x 128(N), 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64
// accumulator VGPRs).  K advances in 128-byte channel runs (64 bf16 / 32 f32) of one tap, so every staged row is one
// fully coalesced 128-B segment of an NHWC pixel.  Global -> registers -> LDS staging, double-buffered (load tile t+1
// while the matrix cores chew tile t, one barrier per tile); halo / parity / tail rows are zero-filled in registers.
// LDS rows are 128 B; 16-B slot s of row r lives at slot s ^ ((r >> 1) & 7), which makes both the ds_write_b128 of
// the staging pass and the ds_read_b128 fragment reads conflict-free.  The epilogue goes through LDS so that each
// lane stores 16 contiguous bytes.  bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32 (exact fp32).
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
// explicit global-address-space pointers: loads through them are global_load (never flat_load, which would force a
// vmcnt(0) in front of every LDS access and serialise the pipeline)
typedef const __attribute__((address_space(1))) char* gptr_t;
typedef const __attribute__((address_space(1))) u32x4* gvec_t;
typedef __attribute__((address_space(3))) char* lds_t;

struct ConvParams {
    int B, IH, IW, Cx, Ck, OH, OW, N, ldy;
    int kh, kw, stride, pad, in_up, pixel_shuffle, per_sample;
    long long x_bstride, w_bstride, y_bstride;     // elements
    int Mtot, n_chunks, n_iters, m_tiles, n_tiles;
    ActEpilogue act;
};

constexpr int BM = 128, BN = 128, ROWB = 128;                  // tile rows / cols, bytes per staged row
constexpr int STAGE_BYTES = (BM + BN) * ROWB;                  // 32 KiB
// Rows that must contribute zeros (halo, parity holes, tile tails) read from here: no branch, no select.
__device__ __attribute__((aligned(256))) unsigned int g_zero_page[16384];   // 64 KiB

__device__ __forceinline__ int swz(int row, int slot) { return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4); }

// LEAN (short K sweeps, n_iters <= ~2: the 1x1 convs on <= 128 channels and the gathered 6-channel input conv, whose
// time is the OUTPUT write, not the contraction): one LDS stage instead of two (32 KiB) and <= 128 VGPRs, so four
// workgroups share a CU and one workgroup's store burst overlaps the others' loads -- latency is hidden by occupancy
// instead of by the in-workgroup double buffer a two-step sweep cannot fill anyway.  LEAN = workgroups per CU the
// register budget is set for (3: 168 VGPRs, 17 spilled outside the loop; 4: 128 VGPRs spills 114 and is slower).
// Measured (bf16, B=16, same box): 1x1 3->512 @256^2 474 -> 390 us, 1x1 128->256 @256^2 346 -> 331 us, 1x1 256->128
// @256^2 255 -> 237 us; whole training step -1 ms.  What these launches still pay (ablations on 1x1 128->256: full 322
// us, no global stores 246, no epilogue 172): the accumulator -> LDS -> 16-B-store epilogue is NOT hidden by the other
// workgroups of the CU (de-phasing them with a start-up delay changed nothing, non-temporal stores neither).
// SPLIT = 3 (T = float; MSG_F32_SPLIT): fp32 storage, every product as bf16 MFMA products on splits of the
// fp32 operands, fp32 accumulation (include/msg_hip.h).  The split happens ONCE PER ELEMENT, on the way from the staging
// registers into LDS: a thread turns the four floats of a staged 16-byte vector into SPLIT planes of four bf16 each (hi = RNE(x),
// mid = RNE(x - hi), lo = RNE(x - hi - mid)) and parks 8 bytes per plane; the LDS image of a K-step is then SPLIT bf16 planes of
// 32 channels per row (row = SPLIT * 64 B + 16 B of padding: conflict-free ds_read_b128), and the MFMA loop is fragment reads
// and v_mfma_f32_32x32x16_bf16 only -- hi hi + hi mid + mid hi + hi lo + lo hi + mid mid, everything
// down to 2^-16 of the leading product.  (The first version split the fragments inside the MFMA loop, every wave its own copy
// of every operand: 5.5 vector instructions per element and sub-step made the six-product kernel 1.13x the speed of the exact
// fp32 MFMA instead of the 2.7x its matrix time allows.)  Register staging only (LDS-DMA cannot convert in flight).
// four floats -> SPLIT x (four bf16 = 8 bytes)
template <int SPLIT>
__device__ __forceinline__ void split_park(const u32x4& raw, char* dst) {
    uint2 pl[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float x0 = __uint_as_float(raw[2 * h]), x1 = __uint_as_float(raw[2 * h + 1]);
        const unsigned hp = (unsigned)f2bf(x0) | ((unsigned)f2bf(x1) << 16);
        const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xffff0000u);
        const unsigned mp = (unsigned)f2bf(r0) | ((unsigned)f2bf(r1) << 16);
        (h ? pl[0].y : pl[0].x) = hp;
        (h ? pl[1].y : pl[1].x) = mp;
        if constexpr (SPLIT == 3) {
            const float q0 = r0 - __uint_as_float(mp << 16), q1 = r1 - __uint_as_float(mp & 0xffff0000u);
            (h ? pl[2].y : pl[2].x) = (unsigned)f2bf(q0) | ((unsigned)f2bf(q1) << 16);
        }
    }
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(dst + p * 64) = pl[p];
}

template <typename T, bool DMA, int LEAN = 0, int SPLIT = 0>
__global__ __launch_bounds__(256, LEAN ? LEAN : 2) void conv_fprop_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                            T* __restrict__ y, const float* __restrict__ bias,
                                                            ConvParams p) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BKE = ROWB / sizeof(T);
    static_assert(!LEAN || (!DMA && sizeof(T) == 2), "lean variant: bf16, register staging");
    static_assert(SPLIT == 0 || sizeof(T) == 4, "split-bf16 products are a mode of the fp32-storage kernel");
    static_assert(SPLIT == 0 || !DMA, "the split planes are written from the staging registers");
    // (SPLIT: rows of SPLIT bf16 planes x 64 B + 16 B of padding -- 144 / 208 B: rows 0..7 of a ds_read_b128 lane group then
    //  start 36 / 52 banks apart, all different 4-bank groups -- 72 KiB / 104 KiB for the two stages)
    constexpr int SROW = SPLIT ? SPLIT * 64 + 16 : ROWB;
    constexpr int STAGE = SPLIT ? (BM + BN) * SROW : STAGE_BYTES;
    // ONE_STAGE: the three-plane image of a K-step is 52 KiB; two stages would leave room for one workgroup per CU (one wave per
    // SIMD, nothing to overlap a barrier with: measured 673 ms per iteration against 618 for the in-loop split).  One stage
    // and two barriers per K-step instead (the LEAN pipeline), 64 KiB with the epilogue's patches: two workgroups per CU.
    constexpr bool ONE_STAGE = LEAN != 0 || SPLIT == 3;
    constexpr int SMEM_BYTES = ONE_STAGE ? (STAGE > 65536 || sizeof(T) == 2 ? STAGE : 65536) : 2 * STAGE;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (int)(L % p.n_tiles) * BN;
    const int m0 = (int)(L / p.n_tiles) * BM;
    const int bz = blockIdx.z;

    // ---- staging assignment: this thread moves 16-B slot `slot` of rows rbase + 32 j (j = 0..3) of A and of B.
    // All address arithmetic is hoisted: per TAP each row gets one pointer (or the zero page when the tap falls into
    // the halo / a parity hole / past the image), per K-step the pointers just advance 128 B.  Rows that must read
    // zeros point into g_zero_page, so the loads need neither a branch nor a select.
    // register staging: thread moves slot (tid&7) of rows (tid>>3) + 32 j.
    // LDS-DMA staging (DMA): one wave-instruction writes 1 KiB = 8 consecutive rows in lane order, so wave w moves rows
    // 32 w + 8 j + (lane>>3), lane&7 is the PHYSICAL slot and the lane fetches the logical slot that belongs there
    // (swizzle applied on the source address; the LDS image is the same as with register staging).
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int slot_phys = tid & 7;
    int row_of[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) row_of[j] = DMA ? (wid * 32 + 8 * j + (lane >> 3)) : ((tid >> 3) + 32 * j);
    const int slot = DMA ? 0 : slot_phys;          // (DMA: per-row logical slot, see slot_j below)
    int a_ih0[4], a_iw0[4];
    long long a_boff[4];
    int a_slot[4];
    bool a_ok[4];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + row_of[j];
        const int slot_j = DMA ? (slot_phys ^ ((row_of[j] >> 1) & 7)) : slot_phys;
        a_ok[j] = m < p.Mtot;
        const int mm = a_ok[j] ? m : 0;
        const int b = p.per_sample ? bz : mm / ohw;
        const int pix = p.per_sample ? mm : mm - b * ohw;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        a_ih0[j] = oh * p.stride - p.pad;
        a_iw0[j] = ow * p.stride - p.pad;
        // byte offset of tap (0,0) of this row's pixel (may point in front of the image for halo rows: it is only
        // dereferenced for taps that land inside); every other tap is this plus ONE wave-uniform term
        a_boff[j] = ((long long)((DMA && p.per_sample) ? 0 : b) * p.x_bstride + slot_j * VEC +   // (DMA, per-sample: the sample sits in the descriptor's base)
                     ((long long)a_ih0[j] * p.IW + a_iw0[j]) * p.Cx) * (long long)sizeof(T);
        a_slot[j] = slot_j;
    }
    const int taps = p.kh * p.kw;
    const gptr_t xbase = (gptr_t)x;
    const gptr_t zbase = (gptr_t)g_zero_page;
    const long long zoff = slot_phys * 16;
    gptr_t pa[4];
    gptr_t pb[4];
    {
        const gptr_t wb = (gptr_t)w + (p.per_sample ? (long long)bz * p.w_bstride : 0) * (long long)sizeof(T);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + row_of[j];
            const bool ok = n < p.N;
            pb[j] = (ok ? wb : zbase) + (ok ? ((long long)n * taps * p.Ck + a_slot[j] * VEC) * (long long)sizeof(T) : zoff);
            pa[j] = zbase;
        }
    }
    // DMA path: the LDS-DMA pieces are addressed through two buffer descriptors with 32-bit per-lane offsets (msg_dma16; a
    // descriptor-addressed 1-KiB piece costs the texture-address unit about half of one with 64 per-lane 64-bit addresses --
    // conv_fprop_pp.hip).  Rows that must read zeros take an out-of-range offset; the launcher keeps tensors >= 2 GiB on the
    // register-staged instantiation.
    msg_desc_t d_x = {0, 0, 0, 0}, d_w = {0, 0, 0, 0};
    unsigned lds0 = 0;
    int va[4] = {MSG_DMA_OOB, MSG_DMA_OOB, MSG_DMA_OOB, MSG_DMA_OOB}, vb[4] = {MSG_DMA_OOB, MSG_DMA_OOB, MSG_DMA_OOB, MSG_DMA_OOB};
    if constexpr (DMA) {
        d_x = msg_make_desc((const char*)x + (p.per_sample ? (long long)bz * p.x_bstride * (long long)sizeof(T) : 0));
        d_w = msg_make_desc((const char*)w + (p.per_sample ? (long long)bz * p.w_bstride * (long long)sizeof(T) : 0));
        lds0 = (unsigned)(unsigned long long)(lds_t)smem;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + row_of[j];
            vb[j] = n < p.N ? (int)(((long long)n * taps * p.Ck + a_slot[j] * VEC) * (long long)sizeof(T)) : MSG_DMA_OOB;
        }
    }
    const bool ragged = (p.Cx % BKE) != 0;          // channel stride not a whole number of 128-B runs
    int ld_tap = -1, ld_chunk = p.n_chunks - 1;     // cursor of the NEXT K-step to load (advanced before each load)
    int kh_ = 0, kw_ = -1;                          // (kh, kw) of ld_tap, advanced without a division
    int st_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) st_off[j] = swz(row_of[j], a_slot[j]);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    u32x4 ra[4], rb[4];
    int dma_stage = 0;                              // LDS stage the next load_next() fills (DMA path)
    auto load_next = [&]() __attribute__((always_inline)) {     // global -> registers for the next K-step, cursor advances
        if (++ld_chunk == p.n_chunks) {         // next tap: one pointer per row (or the zero page)
            ld_chunk = 0;
            ++ld_tap;
            if (++kw_ == p.kw) { kw_ = 0; ++kh_; }
            if (p.in_up == 1) {                 // ~6 vector instructions per row: range test, 64-bit add, select
                const long long tap_off = ((long long)kh_ * p.IW + kw_) * p.Cx * (long long)sizeof(T);   // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = a_ok[j] & ((unsigned)(a_ih0[j] + kh_) < (unsigned)p.IH) &
                                    ((unsigned)(a_iw0[j] + kw_) < (unsigned)p.IW);
                    pa[j] = (ok ? xbase : zbase) + (ok ? a_boff[j] + tap_off : zoff);
                    if constexpr (DMA) va[j] = ok ? (int)(a_boff[j] + tap_off) : MSG_DMA_OOB;
                }
            } else {                            // transposed strided conv as a gather with parity holes
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ih = a_ih0[j] + kh_, iw = a_iw0[j] + kw_;
                    bool ok = a_ok[j] & (ih >= 0) & (iw >= 0) & (ih % p.in_up == 0) & (iw % p.in_up == 0);
                    const int ihs = ih / p.in_up, iws = iw / p.in_up;
                    ok = ok & (ihs < p.IH) & (iws < p.IW);
                    const long long off = a_boff[j] + ((long long)(ihs - a_ih0[j]) * p.IW + (iws - a_iw0[j])) * p.Cx * (long long)sizeof(T);
                    pa[j] = (ok ? xbase : zbase) + (ok ? off : zoff);
                    if constexpr (DMA) va[j] = ok ? (int)off : MSG_DMA_OOB;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool c_bad = ragged && (ld_chunk * BKE + a_slot[j] * VEC + VEC > p.Cx);
            gptr_t a_addr = pa[j];
            if (c_bad) a_addr = zbase + zoff;
            if constexpr (DMA) {
                // wave-uniform LDS destination: rows 32 w + 8 j .. +7 of the stage being filled, lanes in order
                const unsigned la = lds0 + dma_stage * STAGE + (wid_u * 32 + 8 * j) * ROWB;
                msg_dma16(d_x, la, c_bad ? MSG_DMA_OOB : va[j], ld_chunk * ROWB);
                msg_dma16(d_w, la + BM * ROWB, vb[j], (ld_tap * p.n_chunks + ld_chunk) * ROWB);
            } else {
                ra[j] = *(gvec_t)a_addr;
                rb[j] = *(gvec_t)pb[j];
            }
            pa[j] += ROWB;
            pb[j] += ROWB;
        }
    };
    auto park = [&](int stage) __attribute__((always_inline)) {  // registers -> LDS stage
        char* sa = smem + stage * STAGE;
        if constexpr (SPLIT != 0) {
            char* sb = sa + BM * SROW;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // floats 4 slot .. 4 slot + 3 of the row -> 8 bytes per plane
                split_park<SPLIT>(ra[j], sa + row_of[j] * SROW + slot_phys * 8);
                split_park<SPLIT>(rb[j], sb + row_of[j] * SROW + slot_phys * 8);
            }
            return;
        }
        char* sb = sa + BM * ROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<u32x4*>(sa + st_off[j]) = ra[j];
            *reinterpret_cast<u32x4*>(sb + st_off[j]) = rb[j];
        }
    };
    // the MFMAs of one staged K-step (stage base `sa`: 128 A rows, then 128 B rows)
    auto mma_step = [&](const char* sa) __attribute__((always_inline)) {
        const char* sb = sa + BM * ROWB;
        if constexpr (sizeof(T) == 2) {
            // 4 k-steps of 16; the fragments of step kk+1 are requested before the MFMAs of step kk are issued,
            // and the order is pinned (DS x8, then [MFMA x4, DS x4] ...) so LDS latency hides under the matrix pipe
            bf16v8 fa[2][2], fb[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[0][t] = *reinterpret_cast<const bf16v8*>(sa + swz(wm * 64 + t * 32 + lr, lh));
                fb[0][t] = *reinterpret_cast<const bf16v8*>(sb + swz(wn * 64 + t * 32 + lr, lh));
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fa[(kk + 1) & 1][t] = *reinterpret_cast<const bf16v8*>(sa + swz(wm * 64 + t * 32 + lr, 2 * (kk + 1) + lh));
                        fb[(kk + 1) & 1][t] = *reinterpret_cast<const bf16v8*>(sb + swz(wn * 64 + t * 32 + lr, 2 * (kk + 1) + lh));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);   // D^T: see epilogue
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // DS read x8 (steps 0 and 1)
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // MFMA x4   (step 0)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // DS read x4 (step 2)
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // MFMA x4   (step 1)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // DS read x4 (step 3)
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);     // MFMA x8   (steps 2, 3)
        } else if constexpr (SPLIT != 0) {
            // two sub-steps of 16 channels; lane (lr, lh) reads, per plane, the 8 bf16 at k = 16 s + 8 lh of its rows
            const char* sbs = sa + BM * SROW;
#pragma unroll
            for (int sstep = 0; sstep < 2; ++sstep) {
                bf16v8 af[SPLIT][2], bf[SPLIT][2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int pl = 0; pl < SPLIT; ++pl) {
                        af[pl][t] = *reinterpret_cast<const bf16v8*>(sa + (wm * 64 + t * 32 + lr) * SROW + pl * 64 + (2 * sstep + lh) * 16);
                        bf[pl][t] = *reinterpret_cast<const bf16v8*>(sbs + (wn * 64 + t * 32 + lr) * SROW + pl * 64 + (2 * sstep + lh) * 16);
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // small terms first: plane sums 2 (2^-16 of the leading product), then 1 (2^-8), then hi hi
#pragma unroll
                        for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
                            for (int pa = 0; pa <= order; ++pa)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[order - pa][j], af[pa][i], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            // lane half h owns k = 16h .. 16h+15 of the 32-float run (any k order is fine as long as A and B agree)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 fa[2], fb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[t] = *reinterpret_cast<const f32x4*>(sa + swz(wm * 64 + t * 32 + lr, 4 * lh + q));
                    fb[t] = *reinterpret_cast<const f32x4*>(sb + swz(wn * 64 + t * 32 + lr, 4 * lh + q));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][e], fa[i][e], acc[i][j], 0, 0, 0);
            }
        }
    };
    if constexpr (ONE_STAGE) {
        load_next();
        for (int it = 0; it < p.n_iters; ++it) {
            if (it) __syncthreads();                    // everyone done reading the single stage
            park(0);
            if (it + 1 < p.n_iters) load_next();        // flies under the MFMAs below
            __syncthreads();
            mma_step(smem);
        }
    } else {
        // Software pipeline, ONE barrier per K-step, writes placed AFTER it: at step `it` the registers hold step it+1
        // (loaded during step it-1); they are parked in the other LDS stage right after the barrier, the loads of step
        // it+2 are issued, and only then the MFMAs of step `it` run -- LDS writes and global loads both fly under them.
        if constexpr (DMA) {
            dma_stage = 0;
            load_next();                                // step 0 -> stage 0, asynchronously
        } else {
            load_next();
            park(0);
            if (p.n_iters > 1) load_next();
        }
        for (int it = 0; it < p.n_iters; ++it) {
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step `it`'s LDS-DMA has landed (explicit: the compiler's own wait is alias-based)
            __syncthreads();
            if constexpr (DMA) {
                if (it + 1 < p.n_iters) { dma_stage = (it + 1) & 1; load_next(); }   // lands while the MFMAs below run
            } else {
                if (it + 1 < p.n_iters) park((it + 1) & 1);
                if (it + 2 < p.n_iters) load_next();
            }
            mma_step(smem + (it & 1) * STAGE);
        }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // everyone done with the staging buffers: the epilogue reuses them

    // ---- epilogue: accumulators -> LDS (wave-private 64x64 patch) -> 16-B stores.
    // The MFMAs ran with the operands swapped, so an accumulator block is the TRANSPOSED product: lane (lr, lh) owns
    // pixel 32 i + lr and, per group g = e >> 2, the four CONSECUTIVE channels 32 j + 8 g + 4 lh + (0..3).  They go to
    // LDS as ONE packed write (8 B bf16 / 16 B f32); 4-channel unit u of row r lives at unit u ^ (r & 15), which spreads
    // the 16 rows a lane group writes over all banks.
    constexpr int PITCH = 64 * sizeof(T);     // f32: 4 waves x 64 rows x 256 B = exactly the 64 KiB of staging LDS
    constexpr int UNITB = 4 * sizeof(T);
    char* ep = smem + wid * (64 * PITCH);
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
            for (int i = 0; i < 2; ++i) {
                const int row = i * 32 + lr;
                char* dstl = ep + row * PITCH + ((unit ^ (row & 15)) * UNITB);
                if constexpr (sizeof(T) == 2) {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(acc[i][j][4 * g + 0] + bv[0]) | ((unsigned)f2bf(acc[i][j][4 * g + 1] + bv[1]) << 16);
                    pk.y = (unsigned)f2bf(acc[i][j][4 * g + 2] + bv[2]) | ((unsigned)f2bf(acc[i][j][4 * g + 3] + bv[3]) << 16);
                    *reinterpret_cast<uint2*>(dstl) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(dstl) = f32x4{acc[i][j][4 * g + 0] + bv[0], acc[i][j][4 * g + 1] + bv[1],
                                                            acc[i][j][4 * g + 2] + bv[2], acc[i][j][4 * g + 3] + bv[3]};
                }
            }
        }
    constexpr int LPR = 64 / VEC;                                  // lanes per 64-element row
    constexpr int RPP = 64 / LPR;                                  // rows per pass
    constexpr int NPASS = 64 / RPP;
    const int er = lane / LPR, ec = (lane % LPR) * VEC;
    const int n = n0 + wn * 64 + ec;
    int nn = n, q = 0;
    if (p.pixel_shuffle) { const int oc = p.N >> 2; q = n / oc; nn = n - q * oc; }
    const int lim = p.pixel_shuffle ? (p.N >> 2) - nn : p.N - n;  // valid elements left in this channel run
    const bool n_ok = n < p.N;
    // Row coordinates (sample, oh, ow) of this lane's row of pass 0 by division ONCE; later passes are RPP pixels further
    // along and are reached by stepping.  gp[pass] = index of the OUTPUT pixel the row is stored to (sample-major,
    // pixel-shuffled where asked), -1 for rows past M; the fused stage's bias vector and per-row noise values are
    // fetched in the same sweep, before the barrier.
    int gp[NPASS];
    float a_bias[VEC], a_noise[NPASS];
    {
        const int m_first = min(m0 + wm * 64 + er, p.Mtot - 1);
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
        for (int pass = 0; pass < NPASS; ++pass) {
            const bool ok = m0 + wm * 64 + pass * RPP + er < p.Mtot;
            const int pix = oh * p.OW + ow;
            const int g = p.pixel_shuffle ? ((b * 2 * p.OH + 2 * oh + (q >> 1)) * (2 * p.OW) + 2 * ow + (q & 1))
                                          : b * ohw + pix;
            gp[pass] = ok ? g : -1;
            a_noise[pass] = (want_noise && ok) ? nw * p.act.noise[(long long)(p.act.noise_batch == 1 ? 0 : b) * ohw + pix] : 0.f;
            ow += RPP;
            while (ow >= p.OW) { ow -= p.OW; ++oh; }
            if (!p.per_sample) while (oh >= p.OH) { oh -= p.OH; ++b; }
        }
    }
    __syncthreads();
    // All patch rows are read (and, for the residual merge, all residual vectors requested) BEFORE any is used: one LDS
    // / HBM latency per tile instead of one per pass.
    u32x4 v[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int row = pass * RPP + er;
        if constexpr (sizeof(T) == 2)      // 16-B unit = two 8-B units stored at (unit ^ (row & 15)): halves swap on odd rows
            v[pass] = *reinterpret_cast<const u32x4*>(ep + row * PITCH + (((lane & 7) ^ ((row & 15) >> 1)) << 4));
        else
            v[pass] = *reinterpret_cast<const u32x4*>(ep + row * PITCH + (((lane & 15) ^ (row & 15)) << 4));
    }
    if constexpr (sizeof(T) == 2) {
        if (er & 1) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) v[pass] = u32x4{v[pass][2], v[pass][3], v[pass][0], v[pass][1]};
        }
    }
    if (p.act.enabled == 1) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) v[pass] = act_epilogue_apply<T>(v[pass], a_bias, a_noise[pass], p.act.alpha, p.act.scale);
    } else if (p.act.enabled == 2) {                          // residual merge (never with pixel_shuffle)
        const T* rbase = reinterpret_cast<const T*>(p.act.residual) + (n_ok ? n : 0);
        u32x4 r[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass)
            r[pass] = *reinterpret_cast<const u32x4*>(rbase + (long long)max(gp[pass], 0) * p.act.res_ld);
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) v[pass] = residual_epilogue_apply<T>(v[pass], r[pass], p.act.res_gain);
    }
    T* ybase = y + (p.pixel_shuffle ? nn : n);
    if (n_ok && lim >= VEC) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass)
            if (gp[pass] >= 0) *reinterpret_cast<u32x4*>(ybase + (long long)gp[pass] * p.ldy) = v[pass];
    } else if (n_ok) {                                        // ragged channel tail: element stores
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (gp[pass] < 0) continue;
            T tmp[VEC];
            *reinterpret_cast<u32x4*>(tmp) = v[pass];
            T* dst = ybase + (long long)gp[pass] * p.ldy;
            for (int e = 0; e < lim; ++e) dst[e] = tmp[e];
        }
    }
}

extern "C" int msg_conv2d_fprop_pp_try(const void* x, const void* w, const float* bias, void* y,
                                       int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                       int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                       long long w_batch_stride, const ActEpilogue* act, void* stream);

extern "C" int msg_conv2d_fprop_pp_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                                            int kh, int kw, long long w_batch_stride);
extern "C" int msg_conv2d_fprop_row3_try(const void* x, const void* w, const float* bias, void* y,
                                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                         int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                         long long w_batch_stride, const ActEpilogue* act, void* stream);

extern "C" int msg_conv2d_fprop_row3_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                                              int kh, int kw, long long w_batch_stride);
extern "C" int msg_conv2d_fprop_upconv_try(const void* x, const void* w, const float* bias, void* y,
                                           int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                           int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                           long long w_batch_stride, const ActEpilogue* act, void* stream);
extern "C" int msg_conv2d_fprop_thin_try(const void* x, const void* w, const float* bias, void* y,
                                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                         int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                         long long w_batch_stride, const ActEpilogue* act, void* stream);
extern "C" int msg_conv2d_fprop_thin_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                              int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                              int act_mode);

// Which kernel msg_conv2d_fprop would launch for this problem: 5 = the streaming kernels of conv_thin.hip (1x1, <= 8 channels on
// one side; assuming no fused activation), 3 / 4 = conv_fprop_row3_kernel<4,4> / <2,2> (3x3 'same' convs
// on wide maps, activation tile shared by the horizontal taps; 256x256 / 128x128 tile), 2 = conv_fprop_pp_kernel (256x256 ping-pong),
// 1 = conv_fprop_kernel<T, true> (128x128, LDS-DMA staging), 0 = conv_fprop_kernel<T, false> (register staging).
extern "C" int msg_conv2d_fprop_plan(int dtype, int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                                     int kh, int kw, long long w_batch_stride) {
    if (dtype == MSG_BF16 && msg_conv2d_fprop_thin_eligible(B, IH, IW, Cx, Ck, OH, OW, N, N <= 8 ? 8 : (N + 7) / 8 * 8, kh, kw, 1,
                                                            0, 1, 0, 0))
        return 5;
    if (dtype == MSG_BF16) {
        const int r3 = msg_conv2d_fprop_row3_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride);
        if (r3) return r3 == 1 ? 3 : 4;
    }
    if (dtype == MSG_BF16 && msg_conv2d_fprop_pp_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride)) return 2;
    const int esz = dtype == MSG_BF16 ? 2 : 4;
    const int n_iters = kh * kw * (Ck / (128 / esz));
    static const int variant = msg_tunable("MSG_CONV_VARIANT", 0);
    const bool fits31 = (long long)(w_batch_stride ? 1 : B) * IH * IW * Cx * esz < 0x7ffffff0ll &&
                        (long long)N * kh * kw * Ck * esz < 0x7ffffff0ll;
    return (fits31 && (variant == 1 || (variant == 0 && n_iters >= 8))) ? 1 : 0;
}

static int conv2d_fprop_impl(const void* x, const void* w, const float* bias, void* y, int dtype,
                             int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                             int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                             long long w_batch_stride, const ActEpilogue& act, void* stream);

extern "C" int msg_conv2d_fprop(const void* x, const void* w, const float* bias, void* y, int dtype,
                                int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                long long w_batch_stride, void* stream) {
    ActEpilogue act{};
    return conv2d_fprop_impl(x, w, bias, y, dtype, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, in_up,
                             pixel_shuffle, w_batch_stride, act, stream);
}

extern "C" int msg_conv2d_fprop_act_mask(const void* x, const void* w, void* y, int dtype,
                                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                         int kh, int kw, int stride, int pad, long long w_batch_stride,
                                         const float* act_bias, const float* noise, const float* noise_weight,
                                         int noise_batch, float alpha, float scale, unsigned char* mask, void* stream) {
    if (noise && (!noise_weight || (noise_batch != 1 && noise_batch != B))) return MSG_EINVAL;
    if (mask) {
        // only the row-sharing 3x3 kernel writes the sign bytes: the caller asks msg_conv2d_fprop_plan first
        const int plan = msg_conv2d_fprop_plan(dtype, B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride);
        if (dtype != MSG_BF16 || N % 8 || stride != 1 || pad != 1 || (plan != 3 && plan != 4)) return MSG_EUNSUPPORTED;
    }
    ActEpilogue act{act_bias, noise, noise_weight, noise_batch, 1, alpha, scale, nullptr, 0, 0.f, mask};
    return conv2d_fprop_impl(x, w, nullptr, y, dtype, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, 1, 0,
                             w_batch_stride, act, stream);
}

extern "C" int msg_conv2d_fprop_act(const void* x, const void* w, void* y, int dtype,
                                    int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                    int kh, int kw, int stride, int pad, long long w_batch_stride,
                                    const float* act_bias, const float* noise, const float* noise_weight,
                                    int noise_batch, float alpha, float scale, void* stream) {
    return msg_conv2d_fprop_act_mask(x, w, y, dtype, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, w_batch_stride,
                                     act_bias, noise, noise_weight, noise_batch, alpha, scale, nullptr, stream);
}

extern "C" int msg_conv2d_fprop_residual(const void* x, const void* w, void* y, int dtype,
                                         int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                         int kh, int kw, int stride, int pad, long long w_batch_stride,
                                         const void* residual, int res_ld, float gain, void* stream) {
    if (!residual || res_ld < N || (((uintptr_t)residual) & 15u) || res_ld % (dtype == MSG_BF16 ? 8 : 4)) return MSG_EINVAL;
    ActEpilogue act{};
    act.enabled = 2; act.residual = residual; act.res_ld = res_ld; act.res_gain = gain;
    return conv2d_fprop_impl(x, w, nullptr, y, dtype, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, 1, 0,
                             w_batch_stride, act, stream);
}

extern "C" int msg_bias_act_reduce_launch(const float* part_b, float* grad_bias, int C, long long n_b, const float* part_n,
                                          float* grad_nw, long long n_n, void* stream);

// Partial-sum rows / entries of msg_conv2d_fprop_act_backward for this problem: rows of [N] floats (one per sample, pixel tile
// and wave row of the row-sharing kernel) and noise entries (one per sample, tile and wave); 0 rows = the problem does not go to
// that kernel (no fusion).
static void act_backward_partials(int dtype, int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int kh, int kw,
                                  long long w_batch_stride, long long* rows, long long* entries) {
    *rows = *entries = 0;
    const int plan = msg_conv2d_fprop_plan(dtype, B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride);
    if (dtype != MSG_BF16 || (plan != 3 && plan != 4)) return;
    const int hm = plan == 3 ? 256 : 128;
    const long long mtot = w_batch_stride ? (long long)OH * OW : (long long)B * OH * OW;
    const long long tiles = (mtot / hm) * (w_batch_stride ? B : 1);
    *rows = tiles * 2;
    *entries = tiles * (N / hm) * 4;
}

extern "C" long long msg_conv2d_fprop_act_backward_workspace(int dtype, int B, int IH, int IW, int Cx, int Ck, int OH, int OW,
                                                             int N, int kh, int kw, long long w_batch_stride, int has_noise) {
    long long rows, entries;
    act_backward_partials(dtype, B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride, &rows, &entries);
    return rows ? rows * N + (has_noise ? entries : 0) : 0;
}

// The data gradient of a 3x3 'same' conv whose INPUT was the output of a fused bias (+ noise) + leaky-ReLU stage, with that
// stage's backward in the epilogue (ActEpilogue::enabled == 3): y = (conv(x, w) [+ residual]) * (s > 0 ? scale : scale * alpha)
// and the stage's bias / noise-weight gradients -- the map between the two backward nodes is never written.  s: `sign_mask`
// (bytes of msg_conv2d_fprop_act_mask / msg_upfirdn2d_separable_act_mask in tiles mask_tile_m x mask_tile_n; tile_m 1 or a
// multiple of 64) or `sign_map` (the stage's stored output, bf16, channel pitch sign_ld).  Only the row-sharing kernels
// (msg_conv2d_fprop_plan == 3 or 4) have this epilogue: MSG_EUNSUPPORTED otherwise, as for anything but bf16.
extern "C" int msg_conv2d_fprop_act_backward(const void* x, const void* w, void* y, int dtype,
                                             int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                             int kh, int kw, int stride, int pad, long long w_batch_stride,
                                             const void* residual, int res_ld,
                                             const unsigned char* sign_mask, int mask_tile_m, int mask_tile_n,
                                             const void* sign_map, int sign_ld, float alpha, float scale,
                                             float* grad_bias, const float* noise, int noise_batch, float* grad_noise_weight,
                                             float* ws, long long ws_floats, void* stream) {
    if (B == 0) return MSG_OK;
    if (!x || !w || !y || B < 0) return MSG_EINVAL;
    if (!sign_mask == !sign_map) return MSG_EINVAL;                         // exactly one sign source
    if (stride != 1 || pad != 1 || kh != 3 || kw != 3 || ldy != N) return MSG_EUNSUPPORTED;
    long long rows, entries;
    act_backward_partials(dtype, B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride, &rows, &entries);
    if (!rows) return MSG_EUNSUPPORTED;
    if (sign_mask && (mask_tile_m <= 0 || mask_tile_n <= 0 || mask_tile_n % 128 || N % mask_tile_n || (((uintptr_t)sign_mask) & 15u) ||
                      (mask_tile_m != 1 && mask_tile_m % 64) || ((long long)B * OH * OW) % mask_tile_m))
        return MSG_EINVAL;
    if (sign_map && (sign_ld < N || sign_ld % 8 || (((uintptr_t)sign_map) & 15u))) return MSG_EINVAL;
    if (residual && (res_ld < N || res_ld % 8 || (((uintptr_t)residual) & 15u))) return MSG_EINVAL;
    const bool has_noise = noise && grad_noise_weight;
    if (has_noise && noise_batch != 1 && noise_batch != B) return MSG_EINVAL;
    const long long need_b = grad_bias ? rows * N : 0, need_n = has_noise ? entries : 0;
    if (need_b + need_n > 0 && (!ws || ws_floats < need_b + need_n)) return MSG_EINVAL;
    ActEpilogue act{};
    act.enabled = 3; act.alpha = alpha; act.scale = scale;
    act.residual = residual; act.res_ld = res_ld; act.res_gain = 1.f;
    act.mask = const_cast<unsigned char*>(sign_mask); act.mask_tile_m = mask_tile_m; act.mask_tile_n = mask_tile_n;
    act.sign_src = sign_map; act.sign_ld = sign_ld;
    act.noise = has_noise ? noise : nullptr; act.noise_batch = noise_batch;
    act.part_b = grad_bias ? ws : nullptr;
    act.part_n = has_noise ? ws + need_b : nullptr;
    if (!msg_conv2d_fprop_row3_try(x, w, nullptr, y, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, 1, 0, w_batch_stride,
                                   &act, stream))
        return MSG_EUNSUPPORTED;
    if (MSG_CHECK_LAUNCH() != MSG_OK) return MSG_ELAUNCH;
    return msg_bias_act_reduce_launch(act.part_b, grad_bias, N, rows, act.part_n, grad_noise_weight, entries, stream);
}

static int conv2d_fprop_impl(const void* x, const void* w, const float* bias, void* y, int dtype,
                             int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                             int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                             long long w_batch_stride, const ActEpilogue& act, void* stream) {
    if (B == 0) return MSG_OK;
    if (!x || !w || !y || B < 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || N <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || in_up <= 0 || Cx <= 0 || Ck <= 0 || ldy <= 0)
        return MSG_EINVAL;
    const int split = dtype == MSG_F32_SPLIT ? 3 : 0;      // fp32 storage, bf16 MFMA products (msg_hip.h)
    if (split) dtype = MSG_F32;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int esz = dtype == MSG_BF16 ? 2 : 4, vec = 16 / esz, bke = 128 / esz;
    if (Ck % bke || Cx % vec || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15u)) return MSG_EUNSUPPORTED;
    if (pixel_shuffle && (N % 4 || (N / 4) % vec || ldy % vec)) return MSG_EUNSUPPORTED;
    if (!pixel_shuffle && ldy % vec) return MSG_EUNSUPPORTED;
    if (in_up > 1 && stride != 1) return MSG_EUNSUPPORTED;
    if (dtype == MSG_BF16 &&
        msg_conv2d_fprop_thin_try(x, w, bias, y, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, in_up, pixel_shuffle,
                                  w_batch_stride, &act, stream))
        return MSG_CHECK_LAUNCH();                 // 1x1 convs with <= 8 channels on one side: streaming kernels (conv_thin.hip)
    if (dtype == MSG_BF16 &&
        msg_conv2d_fprop_upconv_try(x, w, bias, y, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, in_up,
                                    pixel_shuffle, w_batch_stride, &act, stream))
        return MSG_CHECK_LAUNCH();                 // the generator's sub-pixel up-convolution: activation-stationary kernel
    if (dtype == MSG_BF16 &&
        msg_conv2d_fprop_row3_try(x, w, bias, y, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, in_up,
                                  pixel_shuffle, w_batch_stride, &act, stream))
        return MSG_CHECK_LAUNCH();                 // 3x3 'same' convs on wide maps: activation tile shared by the three horizontal taps
    if (dtype == MSG_BF16 &&
        msg_conv2d_fprop_pp_try(x, w, bias, y, B, IH, IW, Cx, Ck, OH, OW, N, ldy, kh, kw, stride, pad, in_up,
                                pixel_shuffle, w_batch_stride, &act, stream))
        return MSG_CHECK_LAUNCH();                 // large shapes: 256x256 ping-pong kernel (conv_fprop_pp.hip)
    ConvParams p{};
    p.B = B; p.IH = IH; p.IW = IW; p.Cx = Cx; p.Ck = Ck; p.OH = OH; p.OW = OW; p.N = N; p.ldy = ldy;
    p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.in_up = in_up; p.pixel_shuffle = pixel_shuffle;
    p.per_sample = w_batch_stride != 0;
    p.act = act;
    p.x_bstride = (long long)IH * IW * Cx;
    p.w_bstride = w_batch_stride;
    p.y_bstride = pixel_shuffle ? 4ll * OH * OW * ldy : (long long)OH * OW * ldy;
    const long long mtot = p.per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    if (mtot >= (1ll << 31)) return MSG_EUNSUPPORTED;
    p.Mtot = (int)mtot;
    p.n_chunks = Ck / bke;
    p.n_iters = kh * kw * p.n_chunks;
    if ((long long)(p.n_iters + 1) * ROWB + 128 > 65536) return MSG_EUNSUPPORTED;   // zero page covers a full K sweep
    p.m_tiles = (int)((mtot + BM - 1) / BM);
    p.n_tiles = (N + BN - 1) / BN;
    const long long blocks = (long long)p.m_tiles * p.n_tiles;
    if (blocks >= (1ll << 31)) return MSG_EUNSUPPORTED;
    dim3 grid((unsigned)blocks, 1, p.per_sample ? B : 1);
    hipStream_t s = (hipStream_t)stream;
    static const int variant = msg_tunable("MSG_CONV_VARIANT", 0);
    // staging: LDS-DMA (descriptor-addressed, msg_dma16) for long K sweeps, register staging (two steps in flight) for short ones;
    // MSG_CONV_VARIANT=1 / 2 forces DMA / registers (A/B measurements)
    // (the DMA instantiation addresses through 31-bit buffer offsets: the activations one descriptor spans -- the batch, or one
    //  sample with per-sample weights -- and one weight set)
    const bool fits31 = (long long)(p.per_sample ? 1 : B) * p.x_bstride * esz < 0x7ffffff0ll &&
                        (long long)N * kh * kw * Ck * esz < 0x7ffffff0ll;
    // (>= 8 K-steps since the pieces go through descriptors: 1x1 512->128 @64^2, B = 32: 39.3 -> 34.8 us; below that the register-
    //  staged and lean instantiations stay ahead: 1x1 128->256 @256^2 532 vs 597 us)
    const bool dma = fits31 && (variant == 1 || (variant == 0 && p.n_iters >= 8));
    static const int lean_max = msg_tunable("MSG_CONV_LEAN", 4);                       // MSG_CONV_LEAN=<n>: lean variant for n_iters <= n (0 = never)
    if (dtype == MSG_BF16) {
        if (!dma && p.n_iters <= lean_max) hipLaunchKernelGGL((conv_fprop_kernel<bf16_t, false, 3>), grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, bias, p);
        else if (dma) hipLaunchKernelGGL((conv_fprop_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, bias, p);
        else hipLaunchKernelGGL((conv_fprop_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, bias, p);
    } else if (split == 3) {       // (register staging whatever the K length: the planes are written from the staging registers)
        hipLaunchKernelGGL((conv_fprop_kernel<float, false, 0, 3>), grid, dim3(256), 0, s, (const float*)x, (const float*)w, (float*)y, bias, p);
    } else {
        if (dma) hipLaunchKernelGGL((conv_fprop_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, (const float*)w, (float*)y, bias, p);
        else hipLaunchKernelGGL((conv_fprop_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, (const float*)w, (float*)y, bias, p);
    }
    return MSG_CHECK_LAUNCH();
}
