// a3: the generator's sub-pixel up-convolution (2x2 stride-2 transposed modulated conv, multi_stylegan_generator.py:391-403) as
// an ACTIVATION-STATIONARY contraction for its one shape family: K = 512 input channels, N = 4 * 512 output columns (four
// sub-pixel positions x 512 channels), per-sample weights, pixel-shuffled bf16 output.
//
//     y[b, 2 oh + dy, 2 ow + dx, c] = sum_k x[b, oh, ow, k] * w[b][(2 dy + dx) * O + c][k]
//
// Why its own kernel.  On the ping-pong kernel (conv_fprop_pp.hip) this is the launch furthest below its roofline
// (905-950 us at 512 -> 2048 on 128^2, batch 16 = 580-600 TFLOP/s; floor ~420 us): with K = 512 a 256 x 256 tile is only
// EIGHT K-steps long, so every tile pays a prologue that waits on HBM with an idle MFMA pipe and an epilogue (128 KiB of
// stores) that nothing overlaps -- a third of the tile's time -- and all 256 CUs hit their epilogues together.
// Measured here: 810 us (680 TFLOP/s).  Ablations (same box): without the global stores 624-687 us, with every weight
// fragment from one cache line 807 us -- the 1.07 GB of output costs ~200 us that do NOT hide behind the MFMAs in either
// kernel (the memory counter retires in order: weight loads issued behind a slice's stores cannot be consumed before those
// stores are acknowledged; requesting the next slice's fragments ahead of the epilogue, and deeper activation prefetch,
// both ran out of registers at two waves per SIMD: 152 B of scratch, 880 us).  PMC: waves parked in s_waitcnt 70 % of
// their cycles, MFMA busy 32 %, no LDS bank conflicts, L2 hit rate 92 %.
// Here the short K is the asset: a workgroup keeps its 128 pixels x 512 channels (128 KiB) in LDS for its whole life and
// sweeps ALL 2048 output columns over them.
//   * activations: staged ONCE per workgroup by LDS-DMA (one 1-KiB pixel row per wave instruction, XOR-swizzled);
//   * weights: never in LDS.  Each wave owns 64-column slices and reads its MFMA weight fragments straight from global
//     memory / L2 (the per-sample weight set is 2 MiB and shared by the 128 workgroups of the sample).  The MFMA's k index
//     is permuted so that a lane's four consecutive fragments are ONE contiguous 64-byte run: k = 64 t + 32 lh + 8 u + (0..7)
//     for step (t, u) -- whole 128-byte lines per weight row, no re-reads;
//   * no workgroup barrier after the prologue: the four waves drift apart, so one wave's epilogue (accumulators -> 4-KiB
//     wave-private LDS patch -> 16-byte stores of whole 128-byte channel runs) runs beside the other waves' MFMAs, and the
//     chip's stores spread over time instead of arriving in bursts.
// Accumulators: a wave tile is 128 pixels x 64 columns (4 x 2 blocks of 32 x 32, 128 registers), operands swapped so that a
// lane owns a pixel and four consecutive channels (see conv_fprop_pp.hip's epilogue).  One workgroup of EIGHT waves per CU
// (two per SIMD, <= 256 registers each), all 160 KiB of its LDS: 128 KiB activations + eight 4-KiB epilogue patches.
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) char* gptr_t;
typedef __attribute__((address_space(3))) char* lds_t;

struct UpconvParams {
    int B, H, W, Cx, N, O, ldy;
    long long x_bstride, w_bstride, y_bstride;        // elements
    int m_tiles, xcd_samples;
};

constexpr int UC_M = 128, UC_K = 512, UC_ROW = UC_K * 2;     // pixels per workgroup, contraction length, bytes per LDS row
constexpr int UC_PATCH = 32 * 64 * 2;                        // epilogue patch per wave: 32 pixels x 64 channels bf16

constexpr int UC_WAVES = 8;                                  // two per SIMD: one computes while the other waits on LDS / L2 / its epilogue

__global__ __launch_bounds__(64 * UC_WAVES, 1) void conv_upconv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             bf16_t* __restrict__ y, UpconvParams p) {
    __shared__ __attribute__((aligned(16))) char smem[UC_M * UC_ROW + UC_WAVES * UC_PATCH];        // 160 KiB: the whole CU's LDS
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int lr = lane & 31, lh = lane >> 5;
    // Workgroup -> (sample, pixel tile).  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own
    // 4-MiB L2, and every workgroup streams its sample's whole 2-MiB weight set through that L2.  xcd_samples: XCD x takes the
    // samples x, x + 8, ... one after the other, so that the 32 workgroups resident on an XCD read ONE weight set, in step
    // (with the plain order two samples' sets plus the activation / output streams thrash every L2 and the weights come
    // from beyond it: 4 GB per launch at 512 -> 2048 on 128^2, batch 16).
    int b, tile;
    if (p.xcd_samples) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int j = k / p.m_tiles;
        tile = k - j * p.m_tiles;
        b = xcd + 8 * j;
    } else {
        b = blockIdx.x / p.m_tiles;
        tile = blockIdx.x - b * p.m_tiles;
    }
    b = __builtin_amdgcn_readfirstlane(b);
    tile = __builtin_amdgcn_readfirstlane(tile);
    const int m0 = tile * UC_M;
    const int hw = p.H * p.W;

    // ---- prologue: the 128 x 512 activation tile, one pixel row (1 KiB = 64 slots of 16 B) per wave instruction.  Lane l
    // fills PHYSICAL slot l of its row and fetches the logical slot l ^ (row & 15) for it (rows beyond the map read pixel 0:
    // their results are never stored).
    {
        const gptr_t xb = (gptr_t)(x + (long long)b * p.x_bstride);
#pragma unroll 4
        for (int j = 0; j < UC_M / UC_WAVES; ++j) {
            const int r = wid_u * (UC_M / UC_WAVES) + j;
            const int m = m0 + r < hw ? m0 + r : 0;
            const gptr_t src = xb + ((long long)m * p.Cx + ((lane ^ (r & 15)) << 3)) * 2;
            __builtin_amdgcn_global_load_lds(src, (lds_t)(smem + r * UC_ROW), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const bf16_t* wb = w + (long long)b * p.w_bstride;
    bf16_t* yb = y + (long long)b * p.y_bstride;
    char* patch = smem + UC_M * UC_ROW + wid * UC_PATCH;
    const int swz = lr & 15;                                   // swizzle of this lane's pixel rows (32 i + lr: same low bits)
    const int n_slices = p.N / 64;

    // the waves take adjacent 64-column slices at a time: together they write 512 consecutive channels of each pixel
    for (int slice = wid_u; slice < n_slices; slice += UC_WAVES) {
        const int n0 = slice * 64;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // weight fragments of chunk t: row n0 + 32 j + lr, 64 bytes from k = 64 t + 32 lh; chunk t + 1 is requested while
        // chunk t is multiplied (an L2 round trip ahead)
        const bf16_t* wrow[2] = {wb + (long long)(n0 + lr) * UC_K + 32 * lh, wb + (long long)(n0 + 32 + lr) * UC_K + 32 * lh};
        bf16v8 fb[2][2][4];                                    // [buffer][j][u]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) fb[0][j][u] = *reinterpret_cast<const bf16v8*>(wrow[j] + 8 * u);
        // activation fragments: ONE set of registers, refilled block by block -- fragment i of the next step is requested
        // right behind this step's two MFMAs on block i (they have read it by then) and is due eight MFMAs later
        bf16v8 fa[4];
        auto read_a = [&](int step, int i) __attribute__((always_inline)) {
            const int slot = ((8 * (step >> 2) + 4 * lh + (step & 3)) ^ swz) << 4;
            fa[i] = *reinterpret_cast<const bf16v8*>(smem + (32 * i + lr) * UC_ROW + slot);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) read_a(0, i);
#pragma unroll
        for (int t = 0; t < UC_K / 64; ++t) {
            if (t + 1 < UC_K / 64) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        fb[(t + 1) & 1][j][u] = *reinterpret_cast<const bf16v8*>(wrow[j] + 64 * (t + 1) + 8 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int step = 4 * t + u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[t & 1][j][u], fa[i], acc[i][j], 0, 0, 0);
                    if (step + 1 < UC_K / 16) read_a(step + 1, i);
                }
            }
        }

        // ---- epilogue of the slice: 32 pixels at a time through the wave's patch.  Lane (lr, lh) owns pixel 32 i + lr and, per
        // group g = e >> 2, channels 32 j + 8 g + 4 lh + (0..3): one packed 8-byte write; 8-byte unit u of row r lives at
        // unit u ^ (r & 15).  Read back as 16-byte vectors, 8 lanes per 128-byte pixel row, 8 rows per pass.
        const int q = n0 / p.O, c0 = n0 - q * p.O;             // sub-pixel position (dy, dx) and first channel of the slice
        const int dy = q >> 1, dx = q & 1;
        const int er = lane >> 3, ec = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int unit = 8 * j + 2 * g + lh;
                    uint2 pk;
                    pk.x = (unsigned)f2bf(acc[i][j][4 * g + 0]) | ((unsigned)f2bf(acc[i][j][4 * g + 1]) << 16);
                    pk.y = (unsigned)f2bf(acc[i][j][4 * g + 2]) | ((unsigned)f2bf(acc[i][j][4 * g + 3]) << 16);
                    *reinterpret_cast<uint2*>(patch + lr * 128 + ((unit ^ (lr & 15)) << 3)) = pk;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private patch: the wave's own writes have landed)
            u32x4 v[4];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + er;
                v[pass] = *reinterpret_cast<const u32x4*>(patch + row * 128 + ((ec ^ ((row & 15) >> 1)) << 4));
                if (row & 1) v[pass] = u32x4{v[pass][2], v[pass][3], v[pass][0], v[pass][1]};
            }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = m0 + 32 * i + pass * 8 + er;
                if (m < hw) {
                    const int oh = m / p.W, ow = m - oh * p.W;
                    const long long pix = (long long)(2 * oh + dy) * (2 * p.W) + 2 * ow + dx;
                    *reinterpret_cast<u32x4*>(yb + pix * p.ldy + c0 + ec * 8) = v[pass];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (patch reads done before the next block overwrites it)
        }
    }
}

extern "C" int msg_conv2d_fprop_upconv_eligible(int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int kh, int kw,
                                                int stride, int pad, int in_up, int pixel_shuffle, long long w_batch_stride) {
    static const int enabled = msg_tunable("MSG_CONV_UPCONV", 1);
    if (!enabled || !pixel_shuffle || w_batch_stride == 0 || kh != 1 || kw != 1 || stride != 1 || pad != 0 || in_up != 1)
        return 0;
    if (Cx != UC_K || Ck != UC_K || N % 256 || (N / 4) % 64 || IH != OH || IW != OW) return 0;
    const long long hw = (long long)OH * OW;
    // measured against the ping-pong kernel (batch 16): 128^2 maps 810 vs 905-950 us; 64^2 238 vs 221-239; 32^2 91 vs 62 --
    // the 8-wave workgroups need >= 128 pixel tiles per sample to keep every XCD on one weight set
    if (hw < 16384 || hw >= (1ll << 30)) return 0;
    if (((hw + UC_M - 1) / UC_M) * B < 256) return 0;
    return 1;
}

// Called by msg_conv2d_fprop (conv_fprop.hip) in front of the tile kernels; returns 1 if it launched.
extern "C" int msg_conv2d_fprop_upconv_try(const void* x, const void* w, const float* bias, void* y,
                                           int B, int IH, int IW, int Cx, int Ck, int OH, int OW, int N, int ldy,
                                           int kh, int kw, int stride, int pad, int in_up, int pixel_shuffle,
                                           long long w_batch_stride, const ActEpilogue* act, void* stream) {
    if (bias || (act && act->enabled)) return 0;
    if (!msg_conv2d_fprop_upconv_eligible(B, IH, IW, Cx, Ck, OH, OW, N, kh, kw, stride, pad, in_up, pixel_shuffle,
                                          w_batch_stride))
        return 0;
    if ((long long)B * (((long long)OH * OW + UC_M - 1) / UC_M) >= (1ll << 31) || w_batch_stride < (long long)N * UC_K) return 0;
    UpconvParams p{};
    p.B = B; p.H = OH; p.W = OW; p.Cx = Cx; p.N = N; p.O = N / 4; p.ldy = ldy;
    p.x_bstride = (long long)IH * IW * Cx;
    p.w_bstride = w_batch_stride;
    p.y_bstride = 4ll * OH * OW * ldy;
    p.m_tiles = (int)(((long long)OH * OW + UC_M - 1) / UC_M);
    static const int xcd_order = msg_tunable("MSG_UPCONV_XCD", 1);                               // MSG_UPCONV_XCD=0: plain sample-major order (A/B)
    p.xcd_samples = xcd_order && B % 8 == 0;
    dim3 grid((unsigned)((long long)p.m_tiles * B));
    hipLaunchKernelGGL(conv_upconv_kernel, grid, dim3(64 * UC_WAVES), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, p);
    return 1;
}
