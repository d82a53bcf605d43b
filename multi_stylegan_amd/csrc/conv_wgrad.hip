// a3/a4: weight gradient of the channels-last convolutions on the gfx950 matrix cores.
//
//   GW[(z)][o][tap][i] (+)= sum_{b in slice z} sum_{oh,ow} GY[b, oh, ow, o] * X[b, ih(oh,kh), iw(ow,kw), i]
//
// GEMM view: M = O (output channels of the forward conv), N = I (its input channels), K = pixels; both operands are
// stored pixel-major (channels contiguous), i.e. the contraction index is the slow one -- a "TN" product.  The LDS
// tiles therefore keep the global layout ([pixel][channel], one fully coalesced row per pixel) and the transposition
// happens in the fragment read: bf16 uses ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group reads a
// 4-pixel x 16-channel block and each lane receives its channel's 4 pixels), f32 needs none because an f32 MFMA
// operand is one element per lane.  A 64-B rotation of every LDS row by (pixel & 3) keeps the transposed reads of the
// four pixels of a block on disjoint banks.
// Tiling: workgroup = 128 (o) x 128 (i) of ONE tap, 4 waves as 2x2 (64x64 each, 64 accumulator VGPRs); K advances
// 64 pixels (bf16) / 32 pixels (f32) per step, double-buffered through registers like the forward kernel.  The grid
// is 1-D over (channel tile, tap, K-slice) in an XCD-aware order (see the kernel); a K-slice is a sample or a chunk
// of one (per-sample weights), or -- shared weights -- a chunk of the pixels of ALL samples concatenated (the batch is
// folded into K; the chunk count comes from a wave-quantisation cost model in the launcher).  Output is fp32.  A GW that
// is the sum of several K-slices is NOT accumulated with float atomics (their arrival order made the weight gradients,
// and every training step behind them, differ from run to run): each slice stores its partial tile into its own slab of
// a caller-provided workspace, and wgrad_reduce_kernel adds the slabs in slice order -- a fixed association, bit-identical
// results -- writing GW in the kernel layout or, transposed on the way, in the parameter's own [O][I][tap] layout.
// On power-of-two maps the staging loads are buffer loads with a constant per-lane offset and an SGPR cursor (UNI, see
// the kernel); other maps take the generic incremental addressing.
// pixel_shuffle = 1 is the weight gradient of the generator's 2x2 stride-2 transposed conv: tap (dy,dx) pairs
// X[b,h,w,:] with GY[b, 2h+dy, 2w+dx, :].
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* gptr_t;       // explicit global pointers: global_load, never flat
typedef const __attribute__((address_space(1))) u32x4* gvec_t;
typedef __attribute__((address_space(3))) char* lds_t;
__device__ __attribute__((aligned(256))) unsigned int g_wgrad_zero_page[64];   // rows that must contribute zeros read here

struct WgradParams {
    int B, IH, IW, Cx, I, OH, OW, ldgy, O;
    int OWv, OHv;                                     // logical row width / row count of the K loop: OW x OH, or (uniform rows on maps like
                                                      // 127, 63, 31, 15 wide) the next power of two / the next even count -- the logical pixels
                                                      // outside the map contribute zeros
    int kh, kw, stride, pad, pixel_shuffle;
    int per_sample, chunks_per_sample, pix_per_chunk, split;   // split: this launch writes K-slice slabs (see the reduce)
    int xcd_slices;                                   // 1: one K-slice per XCD (single channel tile), 0: tile-major order
    int nz;                                           // number of K-slices (samples x chunks, or chunks when folded)
    int fold;                                         // 1: shared weights, K runs over the concatenated pixels of ALL samples
    int o_tiles, i_tiles, ldgw;                       // ldgw = padded I of the gradient buffer
    int oi_major;                                     // 1: gw[o][i][tap] (the parameter's own layout), 0: gw[o][tap][ldgw]
    float gain;                                       // multiplies the result (equalized-lr scale of the layer)
    long long gw_zstride;
    long long slab;                                   // floats per slab: O * taps * ldgw
};

constexpr int WT = 128;                              // tile extent in both channel dimensions
template <typename T> struct WgCfg { static constexpr int KP = 8192 / (WT * sizeof(T)) * 2; };  // 64 bf16 / 32 f32
// byte offset of 16-B chunk `ch` of pixel row `r` inside a [KP][128] tile, rows rotated by 64 B * (r & 3)
template <typename T> __device__ __forceinline__ int wg_off(int r, int ch) {
    constexpr int ROW = WT * sizeof(T);
    return r * ROW + (((ch << 4) + ((r & 3) << 6)) & (ROW - 1));
}

// UNI ("uniform rows"): when a K-step of KP pixels never straddles output rows unevenly (OW divides KP or KP divides
// OW -- every power-of-two map), each thread's pixel sits at a FIXED (row, column) displacement from the K-step's first
// pixel.  Addresses then split into a per-thread constant byte offset and a wave-uniform part that lives in SGPRs, and
// the loads become buffer loads (descriptor = sample base, voffset = constant, soffset = scalar): out-of-image rows
// get an out-of-range voffset, for which the hardware returns zeros.  This removes ~170 of the ~200 VALU instructions
// per K-step that the generic incremental addressing costs (the kernel was VALU-bound: 12 VALU per MFMA).
constexpr int BUF_OOB = (int)0x80000000;             // voffset >= num_records: the buffer load returns 0

// SPLIT = 3 (T = float; MSG_F32_SPLIT): products as bf16 MFMAs on splits of the fp32 operands, see
// conv_fprop.hip / msg_hip.h.  As there, the split happens once per element on the way from the staging registers into LDS:
// the K-step's image is SPLIT bf16 planes per operand, each laid out exactly like the bf16 kernel's [pixel][channel] tile
// (256-byte rows, 64-byte rotation), so the fragments come from the same transposing reads.
// four floats -> SPLIT x (four bf16 = 8 bytes), plane p at dst + p * plane_bytes
template <int SPLIT>
__device__ __forceinline__ void wg_split_park(const u32x4& raw, char* dst, int plane_bytes) {
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
    for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(dst + p * plane_bytes) = pl[p];
}

template <typename T, bool DMA, bool UNI, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                            float* __restrict__ gw, float* __restrict__ ws, WgradParams p) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int KP = WgCfg<T>::KP;
    constexpr int ROW = WT * sizeof(T);
    constexpr int CPR = ROW / 16;                    // 16-B chunks per pixel row: 16 (bf16) / 32 (f32)
    constexpr int TILE = KP * ROW;                   // 16 KiB
    constexpr int NLD = TILE / 16 / 256;             // 16-B loads per thread per operand: 4
    static_assert(SPLIT == 0 || (sizeof(T) == 4 && !DMA), "split planes: fp32 storage, written from the staging registers");
    constexpr int PROW = 256, PTILE = KP * PROW;     // SPLIT: one bf16 plane of one operand (KP pixel rows x 128 channels): 8 KiB
    constexpr int SSTAGE = SPLIT ? 2 * SPLIT * PTILE : 2 * TILE;   // bytes per stage (both operands)
    // (SPLIT = 3: ONE stage of 48 KiB and two barriers per K-step -- two stages would leave one workgroup per CU)
    constexpr bool ONE_STAGE = SPLIT == 3;
    __shared__ __attribute__((aligned(16))) char smem[(ONE_STAGE ? 1 : 2) * SSTAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // 1-D grid; hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2).
    //  * one channel tile (O, I <= 128: the 9 taps of a K-slice are the only sharers of its gy / x rows): a whole slice
    //    goes to ONE XCD -- id = 8 * slot + xcd, z = 8 * (slot / taps) + xcd, tap = slot % taps -- so the rows are
    //    fetched into one L2 instead of up to eight (measured +18 % on 3x3 128->128 @256^2).  Slices are padded to a
    //    multiple of 8; the padding workgroups leave at once.
    //  * several tiles: tile index fastest, then tap, then slice.  With 8 | tiles an XCD then owns fixed channel
    //    tiles for all taps and slices, i.e. streams only its share of the gy / x columns (measured faster than the
    //    slice-per-XCD order there: 780 vs 644 TFLOP/s on 3x3 512->512 @256^2).
    const int taps_ = p.kh * p.kw;
    const int G = p.o_tiles * p.i_tiles * taps_;
    int z, tile, tap;
    if (p.xcd_slices) {
        const int slot = blockIdx.x >> 3;
        z = ((slot / G) << 3) + (blockIdx.x & 7);
        const int wgi = slot % G;
        tile = wgi / taps_;
        tap = wgi - tile * taps_;
    } else {
        z = blockIdx.x / G;
        const int wgi = blockIdx.x - z * G;
        tap = wgi / (p.o_tiles * p.i_tiles);
        tile = wgi - tap * (p.o_tiles * p.i_tiles);
    }
    // (integer division runs on the vector ALU even for uniform operands: pin the results back into SGPRs, otherwise
    //  the buffer descriptors derived from them are treated as divergent and every load becomes a waterfall loop)
    z = __builtin_amdgcn_readfirstlane(z);
    tile = __builtin_amdgcn_readfirstlane(tile);
    tap = __builtin_amdgcn_readfirstlane(tap);
    if (z >= p.nz) return;
    const int o0 = (tile / p.i_tiles) * WT, i0 = (tile % p.i_tiles) * WT;
    const int kh_ = tap / p.kw, kw_ = tap - kh_ * p.kw;
    const int b = z / p.chunks_per_sample, chunk = z - b * p.chunks_per_sample;
    const int owl = p.OWv;                            // logical row width (== p.OW unless the rows are padded, uniform-row addressing only)
    const int ohl = p.OHv;                            // logical rows per sample (== p.OH unless padded)
    const int npix = ohl * owl;
    const int pix0 = chunk * p.pix_per_chunk;
    const int pix1 = min(p.fold ? p.B * npix : npix, pix0 + p.pix_per_chunk);
    const int n_iters = (pix1 - pix0 + KP - 1) / KP;

    // ---- staging: thread moves 16-B chunk `ch` of pixel rows r0 + RSTEP*j (j = 0..NLD-1) of both operands.  Pixel
    // coordinates advance incrementally (no division in the loop); rows that must read zeros use the zero page.
    constexpr int RSTEP = 256 / CPR;
    // LDS-DMA (DMA): a wave-instruction fills 1 KiB = RPW whole pixel rows in lane order; wave w takes rows
    // w*RPW + 4*RPW*j + lane/CPR, (lane % CPR) is the PHYSICAL 16-B chunk and the lane fetches the logical chunk that the
    // 64-B row rotation puts there.  Register staging keeps the original assignment.
    constexpr int RPW = 1024 / ROW;                   // rows per wave-instruction: 4 (bf16) / 2 (f32)
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const int r0 = DMA ? (wid * RPW + lane / CPR) : tid / CPR;
    const int ch_phys = DMA ? (lane % CPR) : (tid % CPR);
    const int ch = DMA ? ((ch_phys - 4 * (r0 & 3)) & (CPR - 1)) : ch_phys;   // (row & 3) is the same for all j: steps of 4*RPW rows
    const gptr_t zsrc = (gptr_t)g_wgrad_zero_page + (tid & 7) * 16;
    // K-step stride in (sample, row, column) units; without folding a step never leaves its sample
    const int step_b = p.fold ? KP / npix : 0;
    const int step_rem = KP - step_b * npix;
    const int step_h = step_rem / owl, step_w = step_rem % owl;
    const int gyw = p.pixel_shuffle ? 2 * p.OW : p.OW, gyh = p.pixel_shuffle ? 2 * p.OH : p.OH;
    const int oc = o0 + ch * VEC, ic = i0 + ch * VEC;
    const bool oc_ok = oc + VEC <= p.ldgy, ic_ok = ic + VEC <= p.Cx;
    const gptr_t gyb = (gptr_t)gy + ((long long)b * gyh * gyw * p.ldgy + oc) * (long long)sizeof(T);
    const gptr_t xb = (gptr_t)x + ((long long)b * p.IH * p.IW * p.Cx + ic) * (long long)sizeof(T);
    int pix[NLD], oh[NLD], ow[NLD], bq[NLD];           // generic addressing: (sample, row, column) of each staged pixel
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        pix[j] = pix0 + r0 + (DMA ? 4 * RPW : RSTEP) * j;
        bq[j] = p.fold ? pix[j] / npix : 0;           // (folded K: the pixel index runs over all samples)
        const int rem = pix[j] - bq[j] * npix;
        oh[j] = rem / owl;
        ow[j] = rem - oh[j] * owl;
    }
    int st_off[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) st_off[j] = wg_off<T>(r0 + (DMA ? 4 * RPW : RSTEP) * j, ch);

    // ---- UNI staging state: thread constants (VGPR) + wave-uniform cursor (SGPR)
    const int u_gs = p.pixel_shuffle ? 2 : 1, u_gkh = p.pixel_shuffle ? kh_ : 0, u_gkw = p.pixel_shuffle ? kw_ : 0;
    const int u_xs = p.pixel_shuffle ? 1 : p.stride;
    const int u_xkh = p.pixel_shuffle ? 0 : kh_ - p.pad, u_xkw = p.pixel_shuffle ? 0 : kw_ - p.pad;
    const int u_L = p.ldgy * (int)sizeof(T), u_C = p.Cx * (int)sizeof(T);
    __amdgpu_buffer_rsrc_t rs_gy, rs_x;
    int voff_gy[NLD], voff_x[NLD], xh_c[NLD], xw_c[NLD], dw_c[NLD];
    int row_s = 0, col_s = 0, b_s = 0, pix_s = pix0;
    const int gy_sample = gyh * gyw * u_L, x_sample = p.IH * p.IW * u_C;      // bytes (host guarantees < 2^31)
    if constexpr (UNI) {
        const char* gbase = (const char*)gy + (long long)b * gyh * gyw * u_L;
        const char* xbase = (const char*)x + (long long)b * p.IH * p.IW * u_C + ((long long)u_xkh * p.IW + u_xkw) * u_C;
        rs_gy = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, BUF_OOB, 0x00020000);
        rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_OOB, 0x00020000);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int r = r0 + RSTEP * j;
            const int db = p.fold ? r / npix : 0;       // (only maps smaller than a K-step put several samples in one)
            const int rr = r - db * npix;
            const int dh = rr / owl, dw = rr - dh * owl;
            dw_c[j] = dw;
            voff_gy[j] = oc_ok ? db * gy_sample + (dh * u_gs * gyw + dw * u_gs) * u_L + oc * (int)sizeof(T) : BUF_OOB;
            voff_x[j] = ic_ok ? db * x_sample + (dh * u_xs * p.IW + dw * u_xs) * u_C + ic * (int)sizeof(T) : BUF_OOB;
            xh_c[j] = dh * u_xs;
            xw_c[j] = dw * u_xs;
        }
        b_s = p.fold ? pix0 / npix : 0;
        const int in_sample = pix0 - b_s * npix;
        row_s = in_sample / owl;
        col_s = in_sample - row_s * owl;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    u32x4 ra[NLD], rb[NLD];
    int dma_stage = 0;
    auto load_next = [&]() __attribute__((always_inline)) {     // global -> registers for the next K-step
        if constexpr (UNI && !DMA) {
            const int rem = pix1 - pix_s;
            // (soffset is an unsigned 32-bit byte offset: the host checked that the whole tensor fits)
            const int so_gy = (int)((unsigned)b_s * (unsigned)gy_sample +
                                    (unsigned)(((row_s * u_gs + u_gkh) * gyw + col_s * u_gs + u_gkw) * u_L));
            const int so_x = (int)((unsigned)b_s * (unsigned)x_sample + (unsigned)((row_s * u_xs * p.IW + col_s * u_xs) * u_C));
            const int xh_s = row_s * u_xs + u_xkh, xw_s = col_s * u_xs + u_xkw;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                // (padded rows: the columns OW..OWv-1 are not pixels of the map)
                const bool pok = (r0 + RSTEP * j < rem) & (col_s + dw_c[j] < p.OW) & (row_s * u_xs + xh_c[j] < p.OH * u_xs);
                const bool xok = pok & ((unsigned)(xh_s + xh_c[j]) < (unsigned)p.IH) &
                                 ((unsigned)(xw_s + xw_c[j]) < (unsigned)p.IW);
                ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_gy, pok ? voff_gy[j] : BUF_OOB, so_gy, 0);
                rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xok ? voff_x[j] : BUF_OOB, so_x, 0);
            }
            pix_s += KP;
            b_s += step_b;
            row_s += step_h;
            col_s += step_w;
            if (col_s >= owl) { col_s -= owl; ++row_s; }
            if (p.fold && row_s >= ohl) { row_s -= ohl; ++b_s; }
            return;
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const bool pok = pix[j] < pix1;
            int gh = oh[j], gwc = ow[j], xh, xw;
            if (p.pixel_shuffle) {                  // (oh,ow) enumerates the LOW-res grid; GY is 2x larger
                gh = 2 * oh[j] + kh_; gwc = 2 * ow[j] + kw_; xh = oh[j]; xw = ow[j];
            } else {
                xh = oh[j] * p.stride + kh_ - p.pad; xw = ow[j] * p.stride + kw_ - p.pad;
            }
            gptr_t ga = gyb + (((long long)bq[j] * gyh + gh) * gyw + gwc) * p.ldgy * (long long)sizeof(T);
            if (!(pok & oc_ok)) ga = zsrc;
            const bool xok = pok & ic_ok & (xh >= 0) & (xw >= 0) & (xh < p.IH) & (xw < p.IW);
            gptr_t xa = xb + (((long long)bq[j] * p.IH + xh) * p.IW + xw) * p.Cx * (long long)sizeof(T);
            if (!xok) xa = zsrc;
            if constexpr (DMA) {
                lds_t la = (lds_t)(smem + dma_stage * 2 * TILE + (wid_u * RPW + 4 * RPW * j) * ROW);
                __builtin_amdgcn_global_load_lds(ga, la, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(xa, la + TILE, 16, 0, 0);
            } else {
                ra[j] = *(gvec_t)ga;
                rb[j] = *(gvec_t)xa;
            }
            pix[j] += KP;
            bq[j] += step_b;
            oh[j] += step_h;
            ow[j] += step_w;
            if (ow[j] >= owl) { ow[j] -= owl; ++oh[j]; }
            if (p.fold && oh[j] >= ohl) { oh[j] -= ohl; ++bq[j]; }
        }
    };
    auto park = [&](int stage) __attribute__((always_inline)) {
        if constexpr (SPLIT != 0) {
            char* st = smem + stage * SSTAGE;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int r = r0 + RSTEP * j;              // channels 4 ch .. 4 ch + 3 of pixel row r -> 8 bytes per plane
                const int off = r * PROW + ((ch * 8 + ((r & 3) << 6)) & (PROW - 1));
                wg_split_park<SPLIT>(ra[j], st + off, PTILE);
                wg_split_park<SPLIT>(rb[j], st + SPLIT * PTILE + off, PTILE);
            }
            return;
        }
        char* sa = smem + stage * 2 * TILE;
        char* sb = sa + TILE;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            *reinterpret_cast<u32x4*>(sa + st_off[j]) = ra[j];
            *reinterpret_cast<u32x4*>(sb + st_off[j]) = rb[j];
        }
    };
    // one barrier per K-step with the LDS writes placed after it (same pipeline as conv_fprop)
    if (n_iters > 0) {
        if constexpr (DMA) {
            dma_stage = 0;
            load_next();
        } else if constexpr (ONE_STAGE) {
            load_next();
        } else {
            load_next();
            park(0);
            if (n_iters > 1) load_next();
        }
    }
    for (int it = 0; it < n_iters; ++it) {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the LDS-DMA of step `it` has landed
        if constexpr (ONE_STAGE) {
            if (it) __syncthreads();                        // everyone done reading the single stage
            park(0);
            if (it + 1 < n_iters) load_next();              // flies under the MFMAs below
            __syncthreads();
        } else {
            __syncthreads();
            if constexpr (DMA) {
                if (it + 1 < n_iters) { dma_stage = (it + 1) & 1; load_next(); }
            } else {
                if (it + 1 < n_iters) park((it + 1) & 1);
                if (it + 2 < n_iters) load_next();
            }
        }
        {
            const char* sa = smem + (ONE_STAGE ? 0 : (it & 1) * SSTAGE);
            const char* sb = sa + TILE;
            if constexpr (sizeof(T) == 2) {
                // transposed fragment: lane = 16 g + 4 q + pq supplies the address of pixel row (kb + q), channels
                // cb + 4 pq ..+3; it receives channel (cb + lane%16) of pixel rows kb .. kb+3.
                const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
                const int kb = 8 * (g >> 1), cb = 16 * (g & 1);
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                auto frag = [&](const char* base, int ks, int col0) __attribute__((always_inline)) {
                    s16x4 part[2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int r = ks * 16 + kb + 4 * half + q;
                        const char* pa = base + r * ROW + (((col0 + cb + 4 * pq) * 2 + ((r & 3) << 6)) & (ROW - 1));
                        part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
                    }
                    return __builtin_bit_cast(bf16v8, (s16x8)__builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7));
                };
                constexpr int NKS = KP / 16;
                bf16v8 fa[2][2], fb[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[0][t] = frag(sa, 0, wm * 64 + t * 32);
                    fb[0][t] = frag(sb, 0, wn * 64 + t * 32);
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {                  // k-steps of 16 pixels, fragments one step ahead
                    if (ks + 1 < NKS) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            fa[(ks + 1) & 1][t] = frag(sa, ks + 1, wm * 64 + t * 32);
                            fb[(ks + 1) & 1][t] = frag(sb, ks + 1, wn * 64 + t * 32);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);    // DS read x16 (steps 0 and 1)
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // step 2
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // step 3
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else if constexpr (SPLIT != 0) {
                // SPLIT bf16 planes per operand, each a [KP][128] bf16 tile: the bf16 kernel's transposing fragment reads
                const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
                const int kb = 8 * (g >> 1), cb = 16 * (g & 1);
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                auto frag = [&](const char* base, int ks, int col0) __attribute__((always_inline)) {
                    s16x4 part[2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int r = ks * 16 + kb + 4 * half + q;
                        const char* pa = base + r * PROW + (((col0 + cb + 4 * pq) * 2 + ((r & 3) << 6)) & (PROW - 1));
                        part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
                    }
                    return __builtin_bit_cast(bf16v8, (s16x8)__builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7));
                };
                const char* sbp = sa + SPLIT * PTILE;
#pragma unroll
                for (int ks = 0; ks < KP / 16; ++ks) {
                    bf16v8 af[SPLIT][2], bfr[SPLIT][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int pl = 0; pl < SPLIT; ++pl) {
                            af[pl][t] = frag(sa + pl * PTILE, ks, wm * 64 + t * 32);
                            bfr[pl][t] = frag(sbp + pl * PTILE, ks, wn * 64 + t * 32);
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            // small terms first: plane sums 2 (2^-16 of the leading product), 1 (2^-8), then hi hi
#pragma unroll
                            for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
                                for (int pa_ = 0; pa_ <= order; ++pa_)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa_][i], bfr[order - pa_][j], acc[i][j], 0, 0, 0);
                        }
                }
            } else {
                const int lr = lane & 31, lh = lane >> 5;
#pragma unroll 4
                for (int ks = 0; ks < KP / 2; ++ks) {              // k-steps of 2 pixels
                    const int r = 2 * ks + lh;
                    const int rot = (r & 3) << 6;
                    float fa[2], fb[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fa[t] = *reinterpret_cast<const float*>(sa + r * ROW + (((wm * 64 + t * 32 + lr) * 4 + rot) & (ROW - 1)));
                        fb[t] = *reinterpret_cast<const float*>(sb + r * ROW + (((wn * 64 + t * 32 + lr) * 4 + rot) & (ROW - 1)));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: fp32, lanes 0..31 = 32 consecutive input channels (128-B runs)
    const int lr = lane & 31, lh = lane >> 5;
    const int taps = p.kh * p.kw;
    // a K-slice of a split sum owns slab z of the workspace (kernel layout, every element written: zeros included)
    float* gz = p.split ? ws + (long long)z * p.slab : gw + (p.per_sample ? (long long)b * p.gw_zstride : 0);
    const bool oi_major = p.oi_major && !p.split;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ic = i0 + wn * 64 + j * 32 + lr;
            if (ic >= p.ldgw) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = o0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (o >= p.O) continue;
                float* dst = oi_major ? gz + ((long long)o * p.I + ic) * taps + tap
                                      : gz + ((long long)o * taps + tap) * p.ldgw + ic;
                if (oi_major && ic >= p.I) continue;
                *dst = acc[i][j][e] * p.gain;
            }
        }
}

// ---- the fixed-order sum of the K-slice slabs -------------------------------------------------------------------
//   out[(n)][o][tap][i] = ((ws[n*chunks + 0] + ws[n*chunks + 1]) + ...)[o][tap][i]        (oi_major: out[(n)][o][i][tap])
// A block = 64 float4 columns x 4 slice groups: wave g adds its contiguous quarter of the slices in slice order (fully
// coalesced 1-KiB reads), the four partial sums meet in LDS and are added in group order.  The association depends on
// `chunks` only, never on timing.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           long long slab, int chunks, int O, int taps, int I, int ldgw,
                                                           int oi_major) {
    __shared__ float4 part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long nvec = slab >> 2;
    const long long v = (long long)blockIdx.x * 64 + lane;
    const int n = blockIdx.y;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (v < nvec) {
        const int per = (chunks + 3) >> 2;
        const int c0 = grp * per, c1 = min(chunks, c0 + per);
        const float4* src = reinterpret_cast<const float4*>(ws + ((long long)n * chunks + c0) * slab) + v;
#pragma unroll 8
        for (int c = c0; c < c1; ++c, src += nvec) {
            const float4 t = *src;
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    }
    part[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || v >= nvec) return;
    float4 t = part[0][lane];
#pragma unroll
    for (int g = 1; g < 4; ++g) { t.x += part[g][lane].x; t.y += part[g][lane].y; t.z += part[g][lane].z; t.w += part[g][lane].w; }
    if (!oi_major) {
        reinterpret_cast<float4*>(out + (long long)n * slab)[v] = t;
        return;
    }
    const int vpr = ldgw >> 2;                                     // float4 per (o, tap) row
    const long long row = v / vpr;
    const int i = (int)(v - row * vpr) * 4;
    const int o = (int)(row / taps), tap = (int)(row - (long long)o * taps);
    float* dst = out + ((long long)n * O + o) * I * taps + (long long)i * taps + tap;
    const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i + k < I) dst[(long long)k * taps] = e[k];
}

// Launches the reduce for `n_out` results of `chunks` slabs each (shared by conv_wgrad_row3.hip).
extern "C" int msg_wgrad_reduce_launch(const float* ws, float* gw, long long slab, int n_out, int chunks, int O, int taps,
                                       int I, int ldgw, int oi_major, void* stream) {
    const long long nvec = slab >> 2;
    const long long blocks = (nvec + 63) / 64;
    if (blocks >= (1ll << 31) || n_out > 65535) return MSG_EUNSUPPORTED;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks, (unsigned)n_out), dim3(256), 0, (hipStream_t)stream,
                       ws, gw, slab, chunks, O, taps, I, ldgw, oi_major);
    return MSG_CHECK_LAUNCH();
}

// conv_wgrad_row3.hip: returns 0 if the geometry is not its own, 1 if it planned / launched, a negative MSG_E* code on
// error.  *need = workspace floats of the launch it would make (0: no split).  plan_only: no launch.
extern "C" int msg_conv2d_wgrad_row3_try(const void* gy, const void* x, float* gw, int dtype,
                                         int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy, int O, int ldgw,
                                         int kh, int kw, int stride, int pad, int pixel_shuffle,
                                         int per_sample, int k_chunks, int oi_major, float gain,
                                         float* ws, long long ws_floats, int plan_only, long long* need, void* stream);

static int wgrad_impl(const void* gy, const void* x, float* gw, int dtype,
                      int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy, int O, int ldgw,
                      int kh, int kw, int stride, int pad, int pixel_shuffle,
                      int per_sample, int k_chunks, int oi_major, float gain,
                      float* ws, long long ws_floats, int plan_only, long long* need, void* stream) {
    *need = 0;
    if (B == 0) return MSG_OK;
    if (B < 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || O <= 0 || I <= 0 || kh <= 0 ||
        kw <= 0 || stride <= 0 || Cx <= 0 || ldgy <= 0 || ldgw < I || ldgw % 4 || k_chunks <= 0)
        return MSG_EINVAL;
    if (!plan_only && (!gy || !x || !gw)) return MSG_EINVAL;
    const int split = dtype == MSG_F32_SPLIT ? 3 : 0;      // fp32 storage, bf16 MFMA products (msg_hip.h)
    if (split) dtype = MSG_F32;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int esz = dtype == MSG_BF16 ? 2 : 4, vec = 16 / esz;
    if (Cx % vec || ldgy % vec) return MSG_EUNSUPPORTED;
    if (!plan_only && (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gw | (uintptr_t)ws) & 15u)) return MSG_EUNSUPPORTED;
    {
        const int r3 = msg_conv2d_wgrad_row3_try(gy, x, gw, dtype, B, IH, IW, Cx, I, OH, OW, ldgy, O, ldgw, kh, kw, stride, pad,
                                                 pixel_shuffle, per_sample, k_chunks, oi_major, gain, ws, ws_floats, plan_only,
                                                 need, stream);
        if (r3 < 0) return r3;
        if (r3) return MSG_OK;                     // kh x 3 'same' convs on wide maps: three taps per workgroup
    }
    WgradParams p{};
    p.B = B; p.IH = IH; p.IW = IW; p.Cx = Cx; p.I = I; p.OH = OH; p.OW = OW; p.ldgy = ldgy; p.O = O;
    p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.pixel_shuffle = pixel_shuffle;
    p.per_sample = per_sample;
    p.chunks_per_sample = k_chunks;
    const int kp = dtype == MSG_BF16 ? 64 : 32;
    // uniform-row addressing: K-steps map onto whole rows / whole row fragments and per-sample tensors fit 31-bit offsets.
    // Maps one or a few columns short of a power of two -- the discriminator's stride-2 convolutions give 127, 63, 31 -- take it
    // too, on rows padded to that power of two: the K loop runs over OH x OWv logical pixels, the padding columns read zeros
    // (<= 1/16 of the MFMA work), and the ~170 vector instructions per K-step of the generic incremental addressing are gone
    // (these weight gradients ran at 480-490 TFLOP/s; MSG_WGRAD_PAD_ROWS=0: off).
    static const int variant = msg_tunable("MSG_CONV_VARIANT", 0);
    static const int pad_rows = msg_tunable("MSG_WGRAD_PAD_ROWS", 1);
    const long long gy_bytes = (long long)(pixel_shuffle ? 4 : 1) * OH * OW * ldgy * esz;
    const long long x_bytes = (long long)IH * IW * Cx * esz;
    int owv = OW, ohv = OH;
    if (pad_rows && !pixel_shuffle && !(kp % OW == 0 || OW % kp == 0)) {
        int v = 1;
        while (v < OW) v <<= 1;
        if ((long long)v * 15 <= (long long)OW * 16) {
            owv = v;
            // (several padded rows per K-step: a whole number of K-steps per sample needs the row count rounded up too)
            if (v < kp && (OH * v) % kp) ohv = ((OH * v + kp - 1) / kp * kp) / v;
        }
    }
    const bool can_fold0 = !per_sample && variant != 3;
    auto uni_ok = [&](int wl) {
        const long long np = (long long)(wl == OW ? OH : ohv) * wl;
        bool u = variant != 2 && variant != 1 && (kp % wl == 0 || wl % kp == 0) && gy_bytes < (1ll << 31) && x_bytes < (1ll << 31);
        // (a folded K loop with uniform-row addressing needs whole samples per K-step or whole K-steps per sample, and
        //  31-bit offsets over the whole batch -- the SGPR cursor counts against the descriptor's range like the per-lane offset:
        //  with a 2^32 limit here, samples that start beyond 2 GiB read zeros (tests/test_hip_conv.py, test_conv_above_two_gib...);
        //  otherwise the folded loop runs on the generic addressing)
        if (can_fold0 && B * np < (1ll << 31) && u &&
            !((np % kp == 0 || kp % np == 0) && B * gy_bytes < 0x7ffffff0ll && B * x_bytes < 0x7ffffff0ll))
            u = false;
        return u;
    };
    bool uni = uni_ok(owv);
    if (!uni && owv != OW) { owv = OW; uni = uni_ok(OW); }        // (padding did not buy uniform rows: the map as it is)
    if (owv == OW) ohv = OH;
    p.OWv = owv;
    p.OHv = ohv;
    const int npix = ohv * owv;                                    // logical pixels per sample
    p.pix_per_chunk = (((npix + k_chunks - 1) / k_chunks + kp - 1) / kp) * kp;
    long long zs = (long long)B * k_chunks;
    const bool can_fold = can_fold0 && (long long)B * npix < (1ll << 31);
    // shared weights: fold the batch into K (one sweep over the concatenated pixels of all samples), so that small maps
    // still give every workgroup a long K loop and the slabs to add shrink from B*chunks to `chunks` per element
    if (can_fold) {
        const long long steps = ((long long)B * npix + kp - 1) / kp;
        const long long tiles = (long long)((O + WT - 1) / WT) * ((I + WT - 1) / WT) * kh * kw;
        // K split: pick the chunk count that minimises (rounds of 512 co-resident workgroups) x (K-steps per workgroup +
        // a fixed prologue/epilogue cost of ~8 steps, + ~MSG_WGRAD_SLAB_COST steps when the sum is split: a workgroup's
        // 64-KiB slab tile is written once and read once by the reduce).  A fixed target count left a quarter of the chip
        // idle in the last round on some layers (768->768 @32^2: 5 chunks = 3.2 rounds, 274 us; 3 chunks = 1.9 rounds, 226 us).
        static const int target_wgs = msg_tunable("MSG_WGRAD_TARGET_WGS", 0);                      // MSG_WGRAD_TARGET_WGS > 0: the old fixed-target rule (A/B)
        static const int slice_tiles = msg_tunable("MSG_WGRAD_SLICE_TILES", 6);                     // MSG_WGRAD_SLICE_TILES: largest channel-tile count that takes the slice-per-XCD order
        static const int slab_cost = msg_tunable("MSG_WGRAD_SLAB_COST", 6);
        const bool sliced = tiles <= (long long)kh * kw * slice_tiles;       // slices are dealt to the XCDs 8 at a time
        long long chunks = 1;
        if (target_wgs > 0) {
            chunks = (target_wgs + tiles - 1) / tiles;
            if (chunks > steps / 4) chunks = steps / 4;
            if (sliced && chunks >= 8) chunks &= ~7ll;
        } else {
            long long best = -1;
            const long long cmax = steps / 4 < 4096 ? steps / 4 : 4096;
            for (long long c = 1; c <= cmax; c += (sliced && c >= 8 ? 8 : 1)) {
                if (sliced && c > 1 && c < 8) continue;
                const long long rounds = (tiles * c + 511) / 512;
                const long long cost = rounds * ((steps + c - 1) / c + 8 + (c > 1 ? slab_cost : 0));
                if (best < 0 || cost < best) { best = cost; chunks = c; }
                if (tiles * c > 8192) break;
            }
        }
        if (chunks < 1) chunks = 1;
        if (chunks > 65535) chunks = 65535;
        p.fold = 1;
        p.pix_per_chunk = (int)(((steps + chunks - 1) / chunks) * kp);
        zs = ((long long)B * npix + p.pix_per_chunk - 1) / p.pix_per_chunk;
        p.chunks_per_sample = (int)zs;                   // (z = chunk; the sample index derived from it is always 0)
    }
    // K-slices per result: a result that is the sum of several slices goes through slabs + the fixed-order reduce
    const long long chunks_per_out = per_sample ? k_chunks : zs;
    const int n_out = per_sample ? B : 1;
    p.split = chunks_per_out > 1;
    p.o_tiles = (O + WT - 1) / WT;
    p.i_tiles = (I + WT - 1) / WT;
    p.ldgw = ldgw;
    p.gw_zstride = oi_major ? (long long)O * I * kh * kw : (long long)O * kh * kw * ldgw;
    p.slab = (long long)O * kh * kw * ldgw;
    p.oi_major = oi_major;
    p.gain = gain;
    p.nz = (int)zs;
    static const int slice_tiles2 = msg_tunable("MSG_WGRAD_SLICE_TILES", 6);
    p.xcd_slices = (p.o_tiles * p.i_tiles <= slice_tiles2 && variant != 4 && (slice_tiles2 == 1 || zs % 8 == 0 || zs >= 64)) ||
                   (variant == 5 && zs % 8 == 0);   // 4 / 5: A/B switches
    const long long nblk = (p.xcd_slices ? ((zs + 7) / 8) * 8 : zs) * p.o_tiles * p.i_tiles * kh * kw;
    if (zs > (1 << 24) || nblk >= (1ll << 31) || chunks_per_out > (1 << 24)) return MSG_EUNSUPPORTED;
    *need = p.split ? zs * p.slab : 0;
    if (plan_only) return MSG_OK;
    if (p.split && (!ws || ws_floats < *need)) return MSG_EINVAL;
    dim3 grid((unsigned)nblk);
    hipStream_t s = (hipStream_t)stream;
    // Register staging keeps two K-steps of loads in flight; measured faster here than LDS-DMA with one step in
    // flight (685 vs 608 TFLOP/s at 3x3 512->512 @256^2): both operands of this kernel stream from beyond L2.
    const bool dma = variant == 1;                   // MSG_CONV_VARIANT=1 forces LDS-DMA staging, 2 the generic addressing (A/B)
    if (dtype == MSG_BF16) {
        if (dma) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, true, false>), grid, dim3(256), 0, s, (const bf16_t*)gy, (const bf16_t*)x, gw, ws, p);
        else if (uni) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, false, true>), grid, dim3(256), 0, s, (const bf16_t*)gy, (const bf16_t*)x, gw, ws, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, false, false>), grid, dim3(256), 0, s, (const bf16_t*)gy, (const bf16_t*)x, gw, ws, p);
    } else if (split == 3) {
        if (uni) hipLaunchKernelGGL((conv_wgrad_kernel<float, false, true, 3>), grid, dim3(256), 0, s, (const float*)gy, (const float*)x, gw, ws, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<float, false, false, 3>), grid, dim3(256), 0, s, (const float*)gy, (const float*)x, gw, ws, p);
    } else {
        if (dma) hipLaunchKernelGGL((conv_wgrad_kernel<float, true, false>), grid, dim3(256), 0, s, (const float*)gy, (const float*)x, gw, ws, p);
        else if (uni) hipLaunchKernelGGL((conv_wgrad_kernel<float, false, true>), grid, dim3(256), 0, s, (const float*)gy, (const float*)x, gw, ws, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<float, false, false>), grid, dim3(256), 0, s, (const float*)gy, (const float*)x, gw, ws, p);
    }
    const int rc = MSG_CHECK_LAUNCH();
    if (rc != MSG_OK || !p.split) return rc;
    return msg_wgrad_reduce_launch(ws, gw, p.slab, n_out, (int)chunks_per_out, O, kh * kw, I, ldgw, oi_major, stream);
}

extern "C" long long msg_conv2d_wgrad_workspace(int dtype, int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy,
                                                int O, int ldgw, int kh, int kw, int stride, int pad, int pixel_shuffle,
                                                int per_sample, int k_chunks) {
    long long need = 0;
    const int rc = wgrad_impl(nullptr, nullptr, nullptr, dtype, B, IH, IW, Cx, I, OH, OW, ldgy, O, ldgw, kh, kw, stride, pad,
                              pixel_shuffle, per_sample, k_chunks, 0, 1.f, nullptr, 0, 1, &need, nullptr);
    return rc == MSG_OK ? need : (long long)rc;
}

extern "C" int msg_conv2d_wgrad(const void* gy, const void* x, float* gw, int dtype,
                                int B, int IH, int IW, int Cx, int I, int OH, int OW, int ldgy, int O, int ldgw,
                                int kh, int kw, int stride, int pad, int pixel_shuffle,
                                int per_sample, int k_chunks, int oi_major, float gain,
                                float* ws, long long ws_floats, void* stream) {
    long long need = 0;
    return wgrad_impl(gy, x, gw, dtype, B, IH, IW, Cx, I, OH, OW, ldgy, O, ldgw, kh, kw, stride, pad, pixel_shuffle,
                      per_sample, k_chunks, oi_major, gain, ws, ws_floats, 0, &need, stream);
}
