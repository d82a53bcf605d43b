// a3/a4: weight gradient of the kh x 3 'same' convolutions on maps at least 64 pixels wide -- all three horizontal taps
// of a kernel row in ONE workgroup (bf16; the 256^2 / 128^2 / 64^2 layers, where most of the weight-gradient time is).
//
//   GW[(z)][o][kh*3 + kw][i] (+)= sum_{b,oh,ow} GY[b, oh, ow, o] * X[b, oh + kh - pad, ow + kw - 1, i],   kw = 0, 1, 2
//
// conv_wgrad.hip gives every tap its own workgroup, so the 64-pixel GY tile and the (shifted) X tile of a K-step are
// staged nine times; global -> LDS staging is what bounds these kernels ("where the conv kernels stand").
// Here a K-step is a 64-pixel segment of ONE image row (the map width is a multiple of 64): the three horizontal taps
// read the same X pixels shifted by one, so the workgroup stages the GY segment once and 66 X pixels (the segment plus
// one neighbour on each side, zeros at the image border) once, and the transposing fragment reads of tap kw simply
// start kw rows further down the LDS tile.  Per MFMA: a third of the staging traffic, two thirds of the LDS reads
// (the GY fragments are shared by the three taps).  Three accumulator sets (3 x 64 registers) mean one workgroup of four
// waves per CU, in the unified VGPR/AGPR file -- the one-wave-per-SIMD regime, with a third of its staging per MFMA.
//
// Same LDS image as conv_wgrad.hip ([pixel][128 channels], rows rotated by 64 B * (pixel & 3), ds_read_b64_tr_b16),
// same buffer-load addressing (per-thread constant offset + wave-uniform SGPR cursor, out-of-range rows read zeros),
// same fp32 epilogue (plain stores into GW[o][tap][i], or into the K-slice's slab when the sum is split: conv_wgrad.hip).
#include "msg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* r3_lds_t;

struct Row3Params {
    int B, H, W, Cx, I, ldgy, O, ldgw;
    int kh, pad;                                      // kernel rows and vertical padding (kw = 3, horizontal padding 1)
    int per_sample, steps_per_chunk, chunks_per_sample, nz, split;
    int o_tiles, i_tiles, oi_major;
    float gain;
    long long gw_zstride, slab;
};

constexpr int R3_ROW = 256;                           // bytes of one pixel row of a 128-channel tile
constexpr int R3_KP = 64;                             // pixels per K-step
constexpr int R3_TA = R3_KP * R3_ROW;                 // 16 KiB: GY segment
constexpr int R3_TB = (R3_KP + 4) * R3_ROW;           // 17 KiB: X segment + neighbours (66 rows used)
constexpr int R3_STAGE = R3_TA + R3_TB;
constexpr int R3_OOB = (int)0x80000000;

// W32: the map is exactly 32 pixels wide -- a K-step is TWO whole image rows; each occupies 34 rows of the X tile (its 32
// pixels between two zero rows: lanes whose source is out of range), tap kw of pixel p of image row q reads X row 34 q + p + kw.
// (Round 1's kernel of this file ran v_mfma_f32_32x32x16_bf16 in compiler-scheduled clusters; `git log` has it.)
// The contraction runs on v_mfma_f32_16x16x32_bf16 with a hand-scheduled K loop (what conv_fprop_row3.hip does, for the same
// reasons: on real data the 16x16x32 form holds the higher clock, and at 16 cycles per MFMA only ~8 cycles of a gap are
// free for other instructions, so they are placed one per gap by hand -- a sched_barrier per MFMA -- instead of in
// clusters).  A K-step of 64 pixels = two sub-steps of 32; per sub-step and wave 48 MFMAs (3 taps x 4 x 4 blocks of
// 16 x 16), 32 transposing fragment reads for the NEXT sub-step in its first 32 gaps.  The workgroup barrier of a step sits
// between its two sub-steps: by then every wave holds the step's last fragments in registers, so behind the barrier
//   * the step's LDS stage is free and receives the pixels of step t + 3 (three stages: two periods to land);
//   * the first fragments of step t + 1 (landed: every wave waited for its own pieces before the barrier) are read beside
//     the second sub-step's MFMAs.
// Accumulators: 3 x 4 x 4 four-register blocks in AGPRs, tied operands of inline-assembly MFMAs (through the builtin the
// compiler re-assigned and copied them around the loop); the compiler still places the s_waitcnt for their inputs.
#ifdef MSG_WGRAD3_STAMPS
// diagnostic build only (tools/wgrad3_stamps.py; never ship or benchmark it): cycle stamps of K-steps 8..9 of every wave
// of the first 256 workgroups
__device__ unsigned long long g_wgrad3_stamps[256 * 4 * 2 * 8];
#define W3_STAMP(k) do { if (it >= 8 && it < 10 && blockIdx.x < 256 && lane == 0) { unsigned long long tt; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory"); \
    g_wgrad3_stamps[((blockIdx.x * 4 + wid) * 2 + (it - 8)) * 8 + (k)] = tt; } } while (0)
extern "C" int msg_wgrad3_debug_read(void* host_dst, int nbytes) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wgrad3_stamps), nbytes) == hipSuccess ? 0 : -1;
}
#else
#define W3_STAMP(k) do {} while (0)
#endif

// Staging: LDS-DMA (buffer_load ... lds, 1 KiB = four pixel rows per wave instruction) into a THREE-stage ring.  (Rounds 2-3
// staged through registers -- nine 16-B loads and nine LDS parks per thread and K-step; in-kernel stamps,
// profiles/r04_wgrad_row3s_stamps.txt: ~380 + ~400 cycles of a step's ~2 500, 1 536 of them MFMA -- and their 36 registers
// were the third stage.  Same box, 3x3 512->512 @256^2, B = 16: 4 050 -> 3 470 us; profiles/r04_wgrad_row3s_dma_ab.txt.)
// A wave issues 8 (wave 0: 9) pieces per K-step and nothing else; the rotation of the LDS image (64 B x (row & 3), what the
// transposing reads want) is applied on the GLOBAL side -- lane L of a piece lands at row L >> 4, 16-B slot L & 15, and
// fetches the channel chunk that the rotation puts there.  Behind the barrier of step t the stage of step t receives step
// t + 3; the wait in front of a barrier leaves the newest step's pieces in flight (vmcnt(8)).  Where the pieces sit in the
// period does not matter (measured: every second gap of the read-free tail of sub-step 1 -- kept; split over the tails of
// two sub-steps -- same; per-wave instances that stagger the waves over 72 gaps -- 2 % slower, the pieces then collide with
// the fragment reads): a piece costs its wave ~45 cycles of MFMA issue wherever it is.
template <bool W32>
__global__ __launch_bounds__(256, 1) void conv_wgrad_row3s_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x,
                                                                  float* __restrict__ gw, float* __restrict__ ws, Row3Params p) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * R3_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid_u >> 1, wn = wid_u & 1;
    const int tiles = p.o_tiles * p.i_tiles;
    const int G = tiles * p.kh;
    int z = blockIdx.x / G;
    const int rem_ = blockIdx.x - z * G;
    int khi = rem_ / tiles;
    int tile = rem_ - khi * tiles;
    z = __builtin_amdgcn_readfirstlane(z);
    khi = __builtin_amdgcn_readfirstlane(khi);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (z >= p.nz) return;
    const int o0 = __builtin_amdgcn_readfirstlane((tile / p.i_tiles) * 128), i0 = __builtin_amdgcn_readfirstlane((tile % p.i_tiles) * 128);
    const int segs = W32 ? 1 : p.W / R3_KP;
    const int steps_per_sample = W32 ? p.H / 2 : p.H * segs;
    int b = 0, s0, s1;
    if (p.per_sample) {
        b = z / p.chunks_per_sample;
        const int chunk = z - b * p.chunks_per_sample;
        s0 = chunk * p.steps_per_chunk;
        s1 = min(steps_per_sample, s0 + p.steps_per_chunk);
    } else {
        s0 = z * p.steps_per_chunk;
        s1 = min(p.B * steps_per_sample, s0 + p.steps_per_chunk);
    }
    b = __builtin_amdgcn_readfirstlane(b);
    s0 = __builtin_amdgcn_readfirstlane(s0);
    s1 = __builtin_amdgcn_readfirstlane(s1);
    const int n_iters = s1 - s0;
    int b_s = s0 / steps_per_sample;
    int row_s = W32 ? (s0 - b_s * steps_per_sample) * 2 : (s0 - b_s * steps_per_sample) / segs;
    int col_s = W32 ? 0 : (s0 - b_s * steps_per_sample - row_s * segs) * R3_KP;
    b_s = __builtin_amdgcn_readfirstlane(b_s);
    row_s = __builtin_amdgcn_readfirstlane(row_s);
    col_s = __builtin_amdgcn_readfirstlane(col_s);

    // ---- staging: piece j of wave w is the 1-KiB chunk c = w + 4 j of an operand's tile (rows 4 c .. 4 c + 3); the X tile
    // has a 17th chunk (rows 64..67: the right neighbour and spare rows), piece 4 of wave 0.  X row r' holds image column
    // col - 1 + r' (W32: image row q = r' / 34 between two zero rows, pixel 32 q + r' % 34 - 1).
    const int lrow = lane >> 4, chl = ((lane & 15) - 4 * lrow) & 15;      // row inside the chunk (= row & 3), logical channel chunk
    const int oc = o0 + chl * 8, ic = i0 + chl * 8;
    const bool oc_ok = oc + 8 <= p.ldgy, ic_ok = ic + 8 <= p.Cx;
    const int u_L = p.ldgy * 2, u_C = p.Cx * 2;
    int voff_gy[4], voff_x[5];
    bool x_q1[5];                                                         // W32: the lane's row belongs to the step's second image row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * (wid_u + 4 * j) + lrow;
        voff_gy[j] = oc_ok ? r * u_L + oc * 2 : R3_OOB;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = 4 * (j < 4 ? wid_u + 4 * j : 16) + lrow;
        if constexpr (!W32) {
            voff_x[j] = (ic_ok && r < R3_KP + 2) ? r * u_C + ic * 2 : R3_OOB;
            x_q1[j] = false;
        } else {
            const int qq = r >= 34 ? 1 : 0, pos = r - 34 * qq;
            voff_x[j] = (ic_ok && pos >= 1 && pos <= 32) ? (32 * qq + pos - 1) * u_C + ic * 2 : R3_OOB;
            x_q1[j] = qq != 0;
        }
    }
    // (columns outside the map: X row 0 of a row's first segment -- column -1 -- and X row 65 of its last one)
    const int voff_x0_left = (wid_u == 0 && lrow == 0) ? R3_OOB : voff_x[0];
    const int voff_x4_right = lrow == 1 ? R3_OOB : voff_x[4];
    const long long sample_gy = (long long)p.H * p.W * u_L, sample_x = (long long)p.H * p.W * u_C;
    const char* gbase = (const char*)gy + (p.per_sample ? (long long)b * sample_gy : 0);
    const char* xbase = (const char*)x + (p.per_sample ? (long long)b * sample_x : 0) +
                        ((long long)(khi - p.pad) * p.W - (W32 ? 0 : 1)) * u_C;
    // (descriptors as four SGPRs for the inline-assembly DMA below: base, no stride, 2^31 records, raw dword format)
    const msg_desc_t d_gy = msg_make_desc(gbase), d_x = msg_make_desc(xbase);
    const unsigned lds0 = (unsigned)(unsigned long long)(r3_lds_t)smem;

    f32x4 acc[3][4][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[k][i][j][e] = 0.f;

    int so_gy = 0, so_x = 0, col_l = 0;
    bool row_ok = false, row_ok1 = false;
    auto load_setup = [&]() __attribute__((always_inline)) {      // cursor -> SGPR offsets of the step being loaded
        const unsigned pixel = ((unsigned)b_s * (unsigned)p.H + (unsigned)row_s) * (unsigned)p.W + (unsigned)col_s;
        so_gy = (int)(pixel * (unsigned)u_L); so_x = (int)(pixel * (unsigned)u_C);
        row_ok = (unsigned)(row_s + khi - p.pad) < (unsigned)p.H;
        row_ok1 = (unsigned)(row_s + 1 + khi - p.pad) < (unsigned)p.H;
        col_l = col_s;
    };
    auto load_advance = [&]() __attribute__((always_inline)) {
        if constexpr (!W32) { col_s += R3_KP; if (col_s == p.W) { col_s = 0; ++row_s; } }
        else row_s += 2;
        if (row_s == p.H) { row_s = 0; ++b_s; }
    };
    // piece q (0..3: GY, 4..8: X) of the step set up by load_setup into `stage`; `live` false: zeros
    auto dma_piece = [&](int q, int stage, bool live) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (q < 4) {
            msg_dma16(d_gy, lds0 + stage * R3_STAGE + (wid_u + 4 * q) * 1024, live ? voff_gy[q] : R3_OOB, so_gy);
        } else {
            const int j = q - 4;
            if (j == 4 && wid_u != 0) return;
            const unsigned la = lds0 + stage * R3_STAGE + R3_TA + (j < 4 ? wid_u + 4 * j : 16) * 1024;
            int off = voff_x[j];
            if (!W32 && j == 0) off = col_l == 0 ? voff_x0_left : off;
            if (!W32 && j == 4) off = col_l + R3_KP == p.W ? voff_x4_right : off;
            const bool ok = live & (W32 ? (x_q1[j] ? row_ok1 : row_ok) : row_ok);
            msg_dma16(d_x, la, ok ? off : R3_OOB, so_x);
        }
#endif
    };
    auto dma_all = [&](int stage, bool live) __attribute__((always_inline)) {
        load_setup();
#pragma unroll
        for (int q = 0; q < 9; ++q) dma_piece(q, stage, live);
        load_advance();
    };
    // prologue: steps 0, 1, 2 into the three stages
    dma_all(0, n_iters > 0);
    dma_all(1, n_iters > 1);
    dma_all(2, n_iters > 2);

    // transposed fragments (ds_read_b64_tr_b16, see conv_wgrad.hip): within each group of 16 lanes, lane 4 q + pq supplies the
    // address of pixel row (8 g + q), channels 4 pq ..+3 of a 16-channel block, and receives channel (lane % 16) of pixel rows
    // 8 g .. 8 g + 3 (second read: + 4) -- the operand layout of the 16x16x32 MFMA (row lane % 16, k = 8 (lane / 16) ..+7).
    // The row a lane addresses is (multiple of 4) + q + tap shift, so the 64-B rotation of its row only depends on
    // (q + shift) & 3: every address is a per-lane constant (one per operand / tap / 16-channel block) plus a compile-time
    // row offset.  (W32: pixels 32..63 of the K-step sit two rows further down the X tile, which also shifts their rotation:
    // a second set of constants for the second sub-step.)
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    constexpr int NPAR = W32 ? 2 : 1;
    int cA[4], cB[NPAR][3][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cA[t] = (8 * g4 + q) * R3_ROW + ((((wm * 64 + t * 16 + 4 * pq) * 2) + ((q & 3) << 6)) & (R3_ROW - 1));
#pragma unroll
        for (int par = 0; par < NPAR; ++par)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                cB[par][k][t] = R3_TA + (8 * g4 + q) * R3_ROW +
                                ((((wn * 64 + t * 16 + 4 * pq) * 2) + (((q + k + 2 * par) & 3) << 6)) & (R3_ROW - 1));
    }
    auto frag = [&](const char* stage_base, int c, int rows) __attribute__((always_inline)) {
        s16x4 part[2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
            part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(stage_base + c + (rows + 4 * half) * R3_ROW));
        return __builtin_bit_cast(bf16v8, (s16x8)__builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    bf16v8 fa[2][4], fb[2][3][4];
    auto read_nth = [&](int f, int s, const char* base) __attribute__((always_inline)) {
        const int a = f == 0 ? 0 : (f >= 5 && f < 8 ? f - 4 : -1);
        if (a >= 0) fa[s][a] = frag(base, cA[a], 32 * s);
        else {
            const int bidx = f < 5 ? f - 1 : f - 4;                 // 0..11
            const int k = bidx / 4, t = bidx % 4;
            fb[s][k][t] = frag(base, cB[W32 ? s : 0][k][t], 32 * s + k + (W32 ? 2 * s : 0));
        }
    };
    auto mfma = [&](int s, int k, int i, int j) __attribute__((always_inline)) {
        f32x4& c = acc[k][i][j];
        const bf16v8& af = fa[s][i];
        const bf16v8& bfr = fb[s][k][j];
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(af), "v"(bfr));
#endif
    };
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");               // step 0 has landed (steps 1 and 2 in flight; wave 0: two of their pieces too)
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 16; ++f) read_nth(f, 0, smem);

    // One K-step, its stage a compile-time constant (the loop is unrolled by three): the stage offset joins the fragment
    // reads' immediate offsets.
    auto k_step = [&](auto stage_tag, int it) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
        const char* sa = smem + ST * R3_STAGE;
        const char* sn = smem + ((ST + 1) % 3) * R3_STAGE;
        const bool live3 = it + 3 < n_iters;
        W3_STAMP(0);
        // sub-step 0: fragments of sub-step 1 in the first 32 gaps
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            mfma(0, m / 16, (m / 4) % 4, m % 4);
            if (m < 32 && (m & 1) == 0) read_nth(m / 2, 1, sa);    // (a fragment = two transposing reads: gaps m and m + 1)
            __builtin_amdgcn_sched_barrier(0);
            if (m == 31) W3_STAMP(1);
        }
        W3_STAMP(2);
        // own fragment reads done; own pieces of step it + 1 landed (those of step it + 2 stay in flight)
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        W3_STAMP(3);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        W3_STAMP(4);
        // sub-step 1: first fragments of step it + 1; the pieces of step it + 3 into the stage this step just left
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            mfma(1, m / 16, (m / 4) % 4, m % 4);
            if (m < 32 && (m & 1) == 0) read_nth(m / 2, 0, sn);
            if (m == 29) load_setup();
            if (m >= 30 && (m & 1) == 0 && m < 48) dma_piece((m - 30) / 2, ST, live3);
            if (m == 47) load_advance();
            __builtin_amdgcn_sched_barrier(0);
            if (m == 29) W3_STAMP(5);
            if (m == 38) W3_STAMP(6);
        }
        W3_STAMP(7);
    };
    int it = 0;
    for (; it + 2 < n_iters; it += 3) {
        k_step(std::integral_constant<int, 0>{}, it);
        k_step(std::integral_constant<int, 1>{}, it + 1);
        k_step(std::integral_constant<int, 2>{}, it + 2);
    }
    if (it < n_iters) k_step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < n_iters) k_step(std::integral_constant<int, 1>{}, it + 1);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");     // (the last inline-assembly MFMAs retire before anything reads them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the dummy pieces of the last steps write zeros into LDS: they land first)

    // ---- epilogue: fp32; lanes 0..15 of a row group = 16 consecutive input channels, one pass per tap and block
    const int l15 = lane & 15;
    const int taps = p.kh * 3;
    float* gz = p.split ? ws + (long long)z * p.slab : gw + (p.per_sample ? (long long)b * p.gw_zstride : 0);
    const bool oi_major = p.oi_major && !p.split;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int tap = khi * 3 + k;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int icn = i0 + wn * 64 + j * 16 + l15;
                if (icn >= p.ldgw) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = o0 + wm * 64 + i * 16 + 4 * g4 + e;
                    if (o >= p.O) continue;
                    float* dst = oi_major ? gz + ((long long)o * p.I + icn) * taps + tap
                                          : gz + ((long long)o * taps + tap) * p.ldgw + icn;
                    if (oi_major && icn >= p.I) continue;
                    *dst = acc[k][i][j][e] * p.gain;
                }
            }
    }
}

extern "C" int msg_wgrad_reduce_launch(const float* ws, float* gw, long long slab, int n_out, int chunks, int O, int taps,
                                       int I, int ldgw, int oi_major, void* stream);

// Called by msg_conv2d_wgrad (conv_wgrad.hip) after its argument checks; returns 1 if the geometry is this file's (planned,
// and launched unless plan_only), 0 if not, a negative MSG_E* code on error.  *need = workspace floats (0: no split).
extern "C" int msg_conv2d_wgrad_row3_try(const void* gy, const void* x, float* gw, int dtype,
                                         int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy, int O, int ldgw,
                                         int kh, int kw, int stride, int pad, int pixel_shuffle,
                                         int per_sample, int k_chunks, int oi_major, float gain,
                                         float* ws, long long ws_floats, int plan_only, long long* need, void* stream) {
    static const int enabled = msg_tunable("MSG_WGRAD_ROW3", 1);
    static const int w32_on = msg_tunable("MSG_WGRAD_ROW3_W32", 1);                         // MSG_WGRAD_ROW3_W32=0: 32-wide maps stay on conv_wgrad_kernel (A/B)
    const bool w32 = w32_on && OW == 32 && OH % 2 == 0;
    if (!enabled || dtype != MSG_BF16 || kw != 3 || stride != 1 || pixel_shuffle || (OW % R3_KP && !w32) || IH != OH ||
        IW != OW || pad != 1 || kh > 3)
        return 0;
    // (the 'same' geometry: horizontal padding 1 is what the shifted-row trick assumes; vertical padding is free)
    const long long gy_bytes = (long long)OH * OW * ldgy * 2, x_bytes = (long long)IH * IW * Cx * 2;
    const long long nb = per_sample ? 1 : B;
    // (31-bit offsets over everything one descriptor spans -- a sample, or the whole batch when it is folded into K: the SGPR
    //  cursor counts against the descriptor's range like the per-lane offset)
    if (nb * gy_bytes >= 0x7ffffff0ll || nb * x_bytes >= 0x7ffffff0ll) return 0;
    Row3Params p{};
    p.B = B; p.H = OH; p.W = OW; p.Cx = Cx; p.I = I; p.ldgy = ldgy; p.O = O; p.ldgw = ldgw;
    p.kh = kh; p.pad = pad;
    p.per_sample = per_sample;
    p.o_tiles = (O + 127) / 128;
    p.i_tiles = (I + 127) / 128;
    p.oi_major = oi_major;
    p.gain = gain;
    p.gw_zstride = oi_major ? (long long)O * I * kh * 3 : (long long)O * kh * 3 * ldgw;
    p.slab = (long long)O * kh * 3 * ldgw;
    const long long steps_per_sample = w32 ? OH / 2 : (long long)OH * (OW / R3_KP);
    const long long tiles = (long long)p.o_tiles * p.i_tiles * kh;
    long long zs, chunks_per_out;
    if (per_sample) {
        // the caller's k_chunks, or -- when it has no opinion (1) -- the same wave-quantisation model as below over this
        // kernel's ONE workgroup per CU: B * tiles workgroups in rounds of 256.  Batch 16 x 512 -> 512 is 768 workgroups = three
        // full rounds; batch 8 (the path-length pass works on half a batch) is 384 = one and a half, i.e. a quarter of the
        // launch on half-empty hardware (round 5: 1 101 TFLOP/s against 1 352 at batch 16); two K-slices per sample make it
        // three full rounds again, for one extra pass over the 9.4-MB slabs (priced at 12 K-steps: with 6 the 64-step sweeps of the 64^2 maps were split too and lost 17 %).
        if (k_chunks <= 1) {
            static const int slab_cost_ps = msg_tunable("MSG_WGRAD3_SLAB_COST_PS", 12);
            long long best = -1;
            for (long long c = 1; c <= 4 && c * 8 <= steps_per_sample; ++c) {
                const long long rounds = ((long long)B * tiles * c + 255) / 256;
                const long long cost = rounds * ((steps_per_sample + c - 1) / c + 3 + (c > 1 ? slab_cost_ps : 0));
                if (best < 0 || cost < best) { best = cost; k_chunks = (int)c; }
            }
        }
        p.chunks_per_sample = k_chunks;
        p.steps_per_chunk = (int)((steps_per_sample + k_chunks - 1) / k_chunks);
        zs = (long long)B * k_chunks;
        chunks_per_out = k_chunks;
    } else {
        // shared weights: the batch is folded into K; K split by the wave-quantisation cost model of conv_wgrad.hip
        // with ONE workgroup per CU (rounds of 256), ~3 steps of fixed cost per workgroup and, when the sum is split, the
        // write + read of a workgroup's three 64-KiB slab tiles (~MSG_WGRAD3_SLAB_COST steps)
        static const int slab_cost = msg_tunable("MSG_WGRAD3_SLAB_COST", 4);
        const long long steps = (long long)B * steps_per_sample;
        long long chunks = 1, best = -1;
        const long long cmax = steps / 4 < 4096 ? steps / 4 : 4096;
        for (long long c = 1; c <= cmax; ++c) {
            const long long rounds = (tiles * c + 255) / 256;
            const long long cost = rounds * ((steps + c - 1) / c + 3 + (c > 1 ? slab_cost : 0));
            if (best < 0 || cost < best) { best = cost; chunks = c; }
            if (tiles * c > 4096) break;
        }
        p.steps_per_chunk = (int)((steps + chunks - 1) / chunks);
        zs = (steps + p.steps_per_chunk - 1) / p.steps_per_chunk;
        p.chunks_per_sample = (int)zs;
        chunks_per_out = zs;
    }
    p.split = chunks_per_out > 1;
    p.nz = (int)zs;
    const long long nblk = zs * tiles;
    if (zs > (1 << 24) || nblk >= (1ll << 31)) return 0;
    *need = p.split ? zs * p.slab : 0;
    if (plan_only) return 1;
    if (p.split && (!ws || ws_floats < *need)) return MSG_EINVAL;
    if (w32)
        hipLaunchKernelGGL(conv_wgrad_row3s_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)gy, (const bf16_t*)x, gw, ws, p);
    else
        hipLaunchKernelGGL(conv_wgrad_row3s_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)gy, (const bf16_t*)x, gw, ws, p);
    if (MSG_CHECK_LAUNCH() != MSG_OK) return MSG_ELAUNCH;
    if (p.split) {
        const int rc = msg_wgrad_reduce_launch(ws, gw, p.slab, per_sample ? B : 1, (int)chunks_per_out, O, kh * 3, I, ldgw,
                                               oi_major, stream);
        if (rc != MSG_OK) return rc;
    }
    return 1;
}
