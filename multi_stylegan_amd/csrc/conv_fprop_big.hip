// a3/a4: implicit-GEMM convolution, 256x256 workgroup tile with 128x128 WAVE tiles (bf16, large layers).
// EXPERIMENTAL / opt-in (MSG_CONV_BIG=1: LDS-DMA staging, =2: register staging); the default large-layer kernel is
// conv_fprop_pp.hip.  Kept because its ablation switches (MSG_BIG_ABL) are how the limits of this family of kernels
// were measured on 3x3 512->512 @256^2, B=16 (TFLOP/s; MFMA peak 2500):
//     full kernel 1058-1100 | no workgroup barrier 1182 | no fragment reads 1083 | B loads only 1243 | A loads only 1399
//     | no staging traffic 1673 | MFMAs only 1703
// i.e. LDS reads, the barrier and the MFMA stream itself cost little; the global->LDS staging costs 37 % however it is
// done (LDS-DMA == register staging), additively in the A and B halves: with ONE wave per SIMD every stalled VMEM issue
// stalls the MFMA stream behind it.  The 8-wave ping-pong kernel (two waves per SIMD, roles swapped) and the 128x128
// kernel (two workgroups per CU, twice the bytes per flop) land on the same ~1050: a rotated per-tile tap order, a
// padded weight pitch, cheaper tap changes and buffer-descriptor addressing did not move it.  The vendor's asm GEMM of
// this shape (MT256x256x64, 16x16x32 MFMAs) reaches 1380; closing that gap needs instruction-level placement of the
// staging instructions, which HIP C++ does not give.
//
// Same GEMM view, LDS image (128-B rows of 64 K-elements, XOR-swizzled 16-B slots) and LDS-DMA staging as
// conv_fprop.hip, but four waves each own a 128x128 output patch (4x4 MFMA 32x32 blocks, 256 accumulator registers in
// the unified VGPR/AGPR file, one wave per SIMD).  Per K-step a wave issues 64 MFMAs against 32 ds_read_b128 and 16
// DMA pieces -- half the LDS read traffic per MFMA of the 64x64 wave tile (conv_fprop.hip) and two thirds of the
// 128x64 one (conv_fprop_pp.hip), which is what the vendor GEMMs of this size use as well (MT256x256x64).  With a
// single wave per SIMD nothing else hides latency, so the K-step is software-pipelined inside the wave: fragment
// reads of sub-step kk+1 and the DMA of K-step it+1 are interleaved between the MFMAs of sub-step kk
// (sched_group_barrier), one workgroup barrier per K-step.
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* lds_t;

struct ConvParamsBig {
    int B, IH, IW, Cx, Ck, OH, OW, N, ldy;
    int kh, kw, stride, pad, in_up, pixel_shuffle, per_sample;
    long long x_bstride, w_bstride, y_bstride;     // elements
    int Mtot, n_chunks, n_iters, m_tiles, n_tiles;
    int w_pitch;                                   // elements between consecutive rows n of the re-laid weights
    int rotate;                                    // 1: per-M-tile rotated tap order
    ActEpilogue act;
};

constexpr int GM = 256, GN = 256, GROW = 128;                  // tile rows / cols, bytes per staged row
constexpr int GSTAGE = (GM + GN) * GROW;                       // 64 KiB

__device__ __forceinline__ int gswz(int row, int slot) { return row * GROW + ((slot ^ ((row >> 1) & 7)) << 4); }

// ABL (timing ablations, wrong results): 1 = no staging traffic in the K loop, 2 = no fragment reads after sub-step 0,
// 3 = no workgroup barrier, 4 = MFMAs only
template <bool REGS, int ABL = 0>      // REGS: global -> registers -> LDS staging instead of LDS-DMA
__global__ __launch_bounds__(256, 1) void conv_fprop_big_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                bf16_t* __restrict__ y, const float* __restrict__ bias,
                                                                ConvParamsBig p) {
    constexpr int VEC = 8, BKE = 64, ESZ = 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * GSTAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid_u >> 1, wn = wid_u & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (int)(L % p.n_tiles) * GN;
    const int m0 = (int)(L / p.n_tiles) * GM;
    const int bz = blockIdx.z;
    const int ohw = p.OH * p.OW;

    // ---- staging (LDS-DMA through BUFFER loads): one wave-instruction fills 1 KiB = 8 consecutive rows in lane order;
    // wave w moves rows 64 w + 8 j + (lane >> 3), j = 0..7, of A and of B.  lane & 7 is the PHYSICAL slot; the lane fetches
    // the logical slot that the swizzle puts there.  Addressing is descriptor + 32-bit per-lane offset + SGPR offset:
    //  * A: the lane offset of a row is recomputed once per TAP (pixel shifted by the tap, or BUF_OOB when the tap falls
    //    into the halo / a parity hole / past the image: the hardware then returns zeros -- no zero page, no select in
    //    the K loop); within the tap the 128-byte channel runs advance through the SGPR offset;
    //  * B: constant lane offset (row n of the re-laid weights), the K position is the SGPR offset.
    // One VGPR per row instead of a 64-bit pointer, and no vector ALU work per K-step.
    constexpr int BUF_OOB = (int)0x80000000;
    const int slot_phys = lane & 7;
    int a_ih0[8], a_iw0[8], a_b32[8], va[8], vb[8];
    unsigned a_okmask = 0;
    const int taps = p.kh * p.kw;
    const char* xb = (const char*)x + (p.per_sample ? (long long)bz * p.x_bstride * ESZ : 0);
    const char* wb = (const char*)w + (p.per_sample ? (long long)bz * p.w_bstride * ESZ : 0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = wid * 64 + 8 * j + (lane >> 3);
        const int sl = slot_phys ^ ((row >> 1) & 7);
        const int m = m0 + row;
        const bool ok = m < p.Mtot;
        a_okmask |= (ok ? 1u : 0u) << j;
        const int mm = ok ? m : 0;
        const int b = p.per_sample ? 0 : mm / ohw;
        const int pix = p.per_sample ? mm : mm - b * ohw;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        a_ih0[j] = oh * p.stride - p.pad;
        a_iw0[j] = ow * p.stride - p.pad;
        // tap-(0,0) offset of the row (may be "negative" for halo pixels: only used when the tap is in range)
        a_b32[j] = (int)(((long long)b * p.x_bstride + sl * VEC) * ESZ) +
                   (p.in_up == 1 ? (a_ih0[j] * p.IW + a_iw0[j]) * p.Cx * ESZ : 0);   // (host: whole tensor < 2 GiB)
        const int n = n0 + row;
        vb[j] = n < p.N ? (int)(((long long)n * p.w_pitch + sl * VEC) * ESZ) : BUF_OOB;
        va[j] = BUF_OOB;
    }
    const bool ragged = (p.Cx % BKE) != 0;
    // K order: taps are visited in a ROTATED order that depends on the M tile, so that the workgroups running side by
    // side on an XCD (consecutive M tiles, same weights) do not all pull the same 32 KiB weight slice out of L2 at the
    // same moment.
    const int rot_tap = p.rotate ? (int)((L / p.n_tiles) % (unsigned)taps) : 0;
    int ld_tap = rot_tap - 1, ld_chunk = p.n_chunks - 1;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // tap cursor (kh_, kw_) advances without a division; for in_up == 1 the row offset of a tap is the row's tap-(0,0)
    // offset plus ONE wave-uniform term, so a tap change costs ~7 vector instructions per row (it used to cost ~60 and
    // sat in front of the MFMAs of every n_chunks-th K-step)
    int kh_ = rot_tap ? (rot_tap - 1) / p.kw : 0, kw_ = rot_tap ? (rot_tap - 1) % p.kw : -1;
    auto next_tap = [&]() __attribute__((always_inline)) {
        ld_chunk = 0;
        if (++ld_tap == taps) { ld_tap = 0; kh_ = 0; kw_ = 0; }
        else if (++kw_ == p.kw) { kw_ = 0; ++kh_; }
        if (p.in_up == 1) {
            const int tap_off = (kh_ * p.IW + kw_) * p.Cx * ESZ;                     // scalar
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = ((a_okmask >> j) & 1u) & ((unsigned)(a_ih0[j] + kh_) < (unsigned)p.IH) &
                                ((unsigned)(a_iw0[j] + kw_) < (unsigned)p.IW);
                va[j] = ok ? a_b32[j] + tap_off : BUF_OOB;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int ih = a_ih0[j] + kh_, iw = a_iw0[j] + kw_;
            bool ok = ((a_okmask >> j) & 1u) & (ih >= 0) & (iw >= 0);
            if (p.in_up > 1) {
                ok = ok & (ih % p.in_up == 0) & (iw % p.in_up == 0);
                ih /= p.in_up; iw /= p.in_up;
            }
            ok = ok & (ih < p.IH) & (iw < p.IW);
            va[j] = ok ? a_b32[j] + (ih * p.IW + iw) * p.Cx * ESZ : BUF_OOB;
        }
    };
    // DMA piece j (A row block and B row block) of K-step `kstep` (whose tap / chunk the cursor holds) into `stage`.
    // `live` = false on the last K-step: the piece is still issued (straight-line code lets the scheduler interleave it
    // with the MFMAs) but out of range, i.e. it writes zeros into the stage nobody reads any more.
    auto dma_piece = [&](int j, int stage, int kstep, bool live) __attribute__((always_inline)) {
        const int row = wid * 64 + 8 * j + (lane >> 3);
        const int sl = slot_phys ^ ((row >> 1) & 7);
        const bool a_zero = !live | (ragged & (ld_chunk * BKE + sl * VEC + VEC > p.Cx));
        lds_t la = (lds_t)(smem + stage * GSTAGE + (wid_u * 64 + 8 * j) * GROW);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, la, 16, a_zero ? BUF_OOB : va[j], ld_chunk * GROW, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, la + GM * GROW, 16, live ? vb[j] : BUF_OOB, (ld_tap * p.n_chunks + ld_chunk) * GROW, 0, 0);
    };

    auto dma_half = [&](int j, int stage, int kstep, bool live, int which) __attribute__((always_inline)) {
        const int row = wid * 64 + 8 * j + (lane >> 3);
        const int sl = slot_phys ^ ((row >> 1) & 7);
        lds_t la = (lds_t)(smem + stage * GSTAGE + (wid_u * 64 + 8 * j) * GROW);
        if (which == 0) {
            const bool a_zero = !live | (ragged & (ld_chunk * BKE + sl * VEC + VEC > p.Cx));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, la, 16, a_zero ? BUF_OOB : va[j], ld_chunk * GROW, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, la + GM * GROW, 16, live ? vb[j] : BUF_OOB, (ld_tap * p.n_chunks + ld_chunk) * GROW, 0, 0);
        }
    };

    // register staging: piece h (0..15: A rows j = h/2 for even h, B rows for odd h) -> rs[h]; parked with ds_write_b128
    u32x4 rs[16];
    auto reg_load = [&](int h, int kstep, bool live) __attribute__((always_inline)) {
        const int j = h >> 1;
        const int row = wid * 64 + 8 * j + (lane >> 3);
        const int sl = slot_phys ^ ((row >> 1) & 7);
        if ((h & 1) == 0) {
            const bool a_zero = !live | (ragged & (ld_chunk * BKE + sl * VEC + VEC > p.Cx));
            rs[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_zero ? BUF_OOB : va[j], ld_chunk * GROW, 0);
        } else {
            rs[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, live ? vb[j] : BUF_OOB, (ld_tap * p.n_chunks + ld_chunk) * GROW, 0);
        }
    };
    auto reg_park = [&](int h, int stage) __attribute__((always_inline)) {
        const int j = h >> 1;
        char* dst = smem + stage * GSTAGE + ((h & 1) ? GM * GROW : 0) + (wid * 64 + 8 * j + (lane >> 3)) * GROW + slot_phys * 16;
        *reinterpret_cast<u32x4*>(dst) = rs[h];
    };

    // prologue: K-step 0 -> stage 0
    if (++ld_chunk == p.n_chunks) next_tap();
    if constexpr (REGS) {
#pragma unroll
        for (int h = 0; h < 16; ++h) reg_load(h, 0, true);
#pragma unroll
        for (int h = 0; h < 16; ++h) reg_park(h, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dma_piece(j, 0, 0, true);
    }

    // fragment addresses: row = w{m,n}*128 + t*32 + lr, slot = 2 kk + lh; ((row >> 1) & 7) only depends on lr
    const int sx = (lr >> 1) & 7;
    const int fa_base = (wm * 128 + lr) * GROW, fb_base = GM * GROW + (wn * 128 + lr) * GROW;

    for (int it = 0; it < p.n_iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA of step `it` has landed (explicit: the compiler's own wait is alias-based)
        if constexpr (ABL != 3 && ABL != 4) __syncthreads();   // all reads of the other stage are done
        const bool more = it + 1 < p.n_iters;
        const int nstage = (it + 1) & 1;
        if (more && ++ld_chunk == p.n_chunks) next_tap();
        const char* sbase = smem + (it & 1) * GSTAGE;
        bf16v8 fa[2][4], fb[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[0][t] = *reinterpret_cast<const bf16v8*>(sbase + fa_base + t * 32 * GROW + ((lh ^ sx) << 4));
            fb[0][t] = *reinterpret_cast<const bf16v8*>(sbase + fb_base + t * 32 * GROW + ((lh ^ sx) << 4));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // The order below is pinned with scheduling fences: four groups of [4 MFMAs, 2 fragment reads of the next
            // sub-step, 1 DMA instruction]; the reads and the DMA issue in the shadow of the MFMAs in front of them.
            const int so = ((2 * (kk + 1) + lh) ^ sx) << 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
                if (kk + 1 < 4) {
                    if constexpr (ABL == 2 || ABL == 4) {
                        fa[(kk + 1) & 1][i] = fa[kk & 1][i];
                        fb[(kk + 1) & 1][i] = fb[kk & 1][i];
                    } else {
                        fa[(kk + 1) & 1][i] = *reinterpret_cast<const bf16v8*>(sbase + fa_base + i * 32 * GROW + so);
                        fb[(kk + 1) & 1][i] = *reinterpret_cast<const bf16v8*>(sbase + fb_base + i * 32 * GROW + so);
                    }
                }
                if constexpr (ABL == 1 || ABL == 4) {
                } else if constexpr (ABL == 5 || ABL == 6) {         // only the B (5) / only the A (6) half of the traffic
                    if ((i & 1) == (ABL == 5 ? 1 : 0)) dma_half(2 * kk + (i >> 1), nstage, it + 1, more, i & 1);
                } else if constexpr (REGS) {
                    // sub-steps 0,1: two global loads per group (16 in all); sub-steps 2,3: two LDS writes per group.
                    // A load has >= 32 MFMAs (~1000 cycles) to land before its register is parked.
                    const int g = (kk & 1) * 4 + i;                  // 0..7
                    if (kk < 2) { reg_load(2 * g, it + 1, more); reg_load(2 * g + 1, it + 1, more); }
                    else { reg_park(2 * g, nstage); reg_park(2 * g + 1, nstage); }
                } else {
                    dma_half(2 * kk + (i >> 1), nstage, it + 1, more, i & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: wave-private 128 x 128 bf16 patch in LDS (32 KiB per wave = all 128 KiB), then 16-B stores
    constexpr int PITCH = 128 * ESZ;
    char* ep = smem + wid * (128 * PITCH);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = j * 32 + lr;
            const int n = n0 + wn * 128 + col;
            const float bv = (bias && n < p.N) ? bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                reinterpret_cast<bf16_t*>(ep + row * PITCH)[col] = f2bf(acc[i][j][e] + bv);
            }
        }
    const int er = lane >> 4, ec = (lane & 15) * VEC;         // 16 lanes per 128-channel row, 4 rows per pass
    float a_bias[VEC], a_noise[32];
    if (p.act.enabled == 1) {
        const int n = n0 + wn * 128 + ec;
#pragma unroll
        for (int e = 0; e < VEC; ++e) a_bias[e] = (p.act.bias && n + e < p.N) ? p.act.bias[n + e] : 0.f;
        const float nw = p.act.noise ? p.act.noise_w[0] : 0.f;
#pragma unroll
        for (int pass = 0; pass < 32; ++pass) {
            const int m = m0 + wm * 128 + pass * 4 + er;
            float nz = 0.f;
            if (p.act.noise && m < p.Mtot) {
                const int b = p.per_sample ? bz : m / ohw;
                const int pix = p.per_sample ? m : m - b * ohw;
                nz = nw * p.act.noise[(long long)(p.act.noise_batch == 1 ? 0 : b) * ohw + pix];
            }
            a_noise[pass] = nz;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 32; ++pass) {
        const int row = pass * 4 + er;
        const int m = m0 + wm * 128 + row;
        const int n = n0 + wn * 128 + ec;
        if (m >= p.Mtot || n >= p.N) continue;
        const int b = p.per_sample ? bz : m / ohw;
        const int pix = p.per_sample ? m : m - b * ohw;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        bf16_t* dst;
        int nn = n;
        if (p.pixel_shuffle) {
            const int oc = p.N >> 2, q = n / oc;
            nn = n - q * oc;
            dst = y + (long long)b * p.y_bstride + ((long long)(2 * oh + (q >> 1)) * (2 * p.OW) + (2 * ow + (q & 1))) * p.ldy + nn;
        } else {
            dst = y + (long long)b * p.y_bstride + ((long long)oh * p.OW + ow) * p.ldy + n;
        }
        const bf16_t* src = reinterpret_cast<const bf16_t*>(ep + row * PITCH) + ec;
        const int lim = p.pixel_shuffle ? (p.N >> 2) - nn : p.N - n;
        u32x4 v = *reinterpret_cast<const u32x4*>(src);
        if (p.act.enabled == 1) v = act_epilogue_apply<bf16_t>(v, a_bias, a_noise[pass], p.act.alpha, p.act.scale);
        else if (p.act.enabled == 2) {                       // residual merge (never with pixel_shuffle)
            const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.act.residual) +
                            ((long long)b * ohw + pix) * p.act.res_ld + n;
            v = residual_epilogue_apply<bf16_t>(v, *reinterpret_cast<const u32x4*>(rp), p.act.res_gain);
        }
        if (lim >= VEC) {
            *reinterpret_cast<u32x4*>(dst) = v;
        } else {
            bf16_t tmp[VEC];
            *reinterpret_cast<u32x4*>(tmp) = v;
            for (int e = 0; e < lim; ++e) dst[e] = tmp[e];
        }
    }
}

extern "C" int msg_conv2d_fprop_pp_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N,
                                            int kh, int kw, long long w_batch_stride);

// Called by msg_conv2d_fprop (conv_fprop.hip) when MSG_CONV_BIG selects this kernel for the large-tile shapes;
// returns 1 if it launched.
extern "C" int msg_conv2d_fprop_big_try(const void* x, const void* w, const float* bias, void* y,
                                        int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                        int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                        long long w_batch_stride, const ActEpilogue* act, void* stream) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("MSG_CONV_BIG"); enabled = e ? atoi(e) : 0; }
    if (!enabled) return 0;
    if (!msg_conv2d_fprop_pp_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, w_batch_stride)) return 0;
    if (pixel_shuffle && ((N >> 2) % 8)) return 0;
    const long long x_bytes = (long long)(w_batch_stride ? 1 : B) * IH * IW * Cx * 2;
    const long long w_bytes = (long long)N * kh * kw * Ck * 2;
    if (x_bytes >= 0x7ffffff0ll || w_bytes >= 0x7ffffff0ll) return 0;    // 31-bit buffer offsets
    const bool per_sample = w_batch_stride != 0;
    const long long mtot = per_sample ? (long long)OH * OW : (long long)B * OH * OW;
    ConvParamsBig p{};
    p.B = B; p.IH = IH; p.IW = IW; p.Cx = Cx; p.Ck = Ck; p.OH = OH; p.OW = OW; p.N = N; p.ldy = ldy;
    p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.in_up = in_up; p.pixel_shuffle = pixel_shuffle;
    p.per_sample = per_sample;
    if (act) p.act = *act;
    p.x_bstride = (long long)IH * IW * Cx;
    p.w_bstride = w_batch_stride;
    p.y_bstride = pixel_shuffle ? 4ll * OH * OW * ldy : (long long)OH * OW * ldy;
    p.Mtot = (int)mtot;
    p.n_chunks = Ck / 64;
    p.n_iters = kh * kw * p.n_chunks;
    static int wpitch_extra = -1;                  // experiment: MSG_BIG_WPITCH = extra elements per weight row
    if (wpitch_extra < 0) { const char* e = getenv("MSG_BIG_WPITCH"); wpitch_extra = e ? atoi(e) : 0; }
    p.w_pitch = kh * kw * Ck + wpitch_extra;
    static int rot = -1;
    if (rot < 0) { const char* e = getenv("MSG_BIG_ROT"); rot = e ? atoi(e) : 0; }
    p.rotate = rot;
    p.m_tiles = (int)((mtot + GM - 1) / GM);
    p.n_tiles = (N + GN - 1) / GN;
    const long long blocks = (long long)p.m_tiles * p.n_tiles;
    dim3 grid((unsigned)blocks, 1, per_sample ? B : 1);
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("MSG_BIG_ABL"); abl = e ? atoi(e) : 0; }
#define BIG_ABL(N_) if (abl == N_) { hipLaunchKernelGGL((conv_fprop_big_kernel<false, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, bias, p); return 1; }
    BIG_ABL(1) BIG_ABL(2) BIG_ABL(3) BIG_ABL(4) BIG_ABL(5) BIG_ABL(6)
#undef BIG_ABL
    if (enabled == 2)
        hipLaunchKernelGGL(conv_fprop_big_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    else
        hipLaunchKernelGGL(conv_fprop_big_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, (bf16_t*)y, bias, p);
    return 1;
}
