// Fused non-local attention of the discriminator's NonLocalBlock:
//     O = softmax(Q K^T) V        Q = theta(x) [Nq, DK], K = maxpool(phi(x)) [Nk, DK], V = maxpool(g(x)) [Nk, DV]
// (reference multi_stylegan/u_net_2d_discriminator.py:359-381: two torch.bmm around an F.softmax, with the
// [B, Nq, Nk] attention map beta -- 16 MB per sample at 256^2, 268 MB at 512^2 -- written to and re-read from HBM and
// kept for backward).  Here beta never leaves the CU: forward keeps only the row statistics L = logsumexp(S), backward
// recomputes the probabilities from Q, K and L.
//
// Design (gfx950, wave64, MFMA 32x32):
//  * every product is an "NT" contraction C[m][n] = sum_k A[m][k] * Bt[n][k] with k contiguous in BOTH operands, so every
//    MFMA fragment is one 16-byte LDS / global read (bf16: 8 elements of v_mfma_f32_32x32x16_bf16; fp32 storage uses the
//    exact v_mfma_f32_32x32x2_f32, one element per lane).  Operands whose natural layout has k strided (V in O = P V, K in
//    dQ = dS K, Q and dO in the key-side gradients) are taken from transposed copies the caller passes (small tensors:
//    [B, N, 48..192]), not gathered with transposing LDS reads.
//  * the score tile is computed TRANSPOSED where the softmax statistics are needed per query (forward, dQ): S^T = K Q^T
//    puts the query on the lane, so max / sum / L / delta are lane-local (one cross-half shuffle), and each lane owns 4
//    consecutive keys per accumulator group, which go to the wave's private LDS scratch as one packed 8-byte store
//    (P[q][key], the A/B operand layout of the next product).
//  * softmax in two sweeps over the keys instead of the online rescale: sweep 1 computes only the row maxima (3 MFMAs per
//    32x32 tile, K = 48), sweep 2 computes P = exp(S - m), l = sum P and accumulates O^T = V^T P^T un-normalised; O is
//    scaled by 1/l once at the end (lane-local: the query is the lane).  +20 % MFMA work in exchange for no accumulator
//    rescaling (96 multiplies per lane per key block) and no data-dependent rescale branch.
//  * backward as two kernels without atomics: dQ per query block (sweeping the keys), dK and dV per key block (sweeping
//    the queries); both recompute S and dP = dO V^T.  When B * Nk / 128 key blocks are too few for the 256 CUs the query
//    sweep is split over several workgroups that write fp32 partial rows, summed by a small third kernel (fixed order:
//    results stay deterministic).
//  * operand blocks go global -> registers -> LDS with the loads of block i+1 issued before block i is multiplied.
// One workgroup = 4 waves x 32 queries (or keys); K / V (or Q / dO) blocks are shared through LDS.
#include "msg_common.h"
#include <stdlib.h>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16v8 Frag;
    static constexpr int KS = 16;      // contraction elements per MFMA
    static constexpr int PAD = 8;      // row padding of LDS tiles, elements (16 bytes)
    static constexpr int VEC = 8;      // elements per 16 bytes
    static __device__ __forceinline__ Frag load(const bf16_t* row, int s, int h) {
        return *reinterpret_cast<const Frag*>(row + 16 * s + 8 * h);
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot(Frag a, Frag b) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)a[j], (float)b[j], acc);
        return acc;
    }
    static __device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
        uint2 v;
        v.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
        v.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
        *reinterpret_cast<uint2*>(p) = v;
    }
    // The fragment load(row (col + r) of the TRANSPOSE, s, h) would give -- column col + (lane & 31) of `tile`, rows
    // 16 s + 8 h .. + 7 -- taken from the row-major tile itself with gfx950's transposing LDS read (ds_read_b64_tr_b16: a
    // 16-lane group reads a 4-row x 16-column block, each lane receives its column's 4 rows; conv_wgrad.hip has the lane
    // algebra).  pitch in elements, a multiple of 4.
    static __device__ __forceinline__ Frag load_t(const bf16_t* tile, int pitch, int col, int s, int lane) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
        const int kb = 8 * (g >> 1), cb = 16 * (g & 1);
        s16x4 part[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bf16_t* pa = tile + (16 * s + kb + 4 * half + q) * pitch + col + cb + 4 * pq;
            part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
        }
        return __builtin_bit_cast(Frag, (s16x8)__builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7));
    }
};
template <> struct Mma<float> {
    typedef float Frag;
    static constexpr int KS = 2;
    static constexpr int PAD = 4;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ Frag load(const float* row, int s, int h) { return row[2 * s + h]; }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot(Frag a, Frag b) { return a * b; }
    static __device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
        *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    }
    static __device__ __forceinline__ Frag load_t(const float* tile, int pitch, int col, int s, int lane) {
        return tile[(2 * s + (lane >> 5)) * pitch + col + (lane & 31)];       // (one element per lane: nothing to transpose)
    }
};

// A ROWS x COLS tile on its way global -> registers -> LDS, 16 bytes per thread and step.  Split in two so that the
// global loads of block i+1 are in flight while block i is being multiplied (one wave per SIMD: nothing else hides them).
// PF = false: load() only remembers the address and store() copies straight through -- no registers held across the
// compute phase (for the kernels that run two waves per SIMD, where the other wave hides the load).
template <typename T, int ROWS, int COLS, bool PF, int THREADS = 256>
struct Stager {
    static constexpr int VEC = 16 / sizeof(T), CPR = COLS / VEC, TOTAL = ROWS * CPR, N = (TOTAL + THREADS - 1) / THREADS;
    static_assert(COLS % VEC == 0, "tile rows are whole 16-byte vectors");
    uint4 r[PF ? N : 1];
    const T* src;
    long long pitch;
    __device__ __forceinline__ void load(const T* g, long long g_pitch, int tid) {
        if constexpr (!PF) { src = g; pitch = g_pitch; return; }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + THREADS * i;
            if (TOTAL % THREADS == 0 || c < TOTAL) {
                const int row = c / CPR, cc = c - row * CPR;
                r[PF ? i : 0] = *reinterpret_cast<const uint4*>(g + row * g_pitch + cc * VEC);
            }
        }
    }
    __device__ __forceinline__ void store(T* lds, int l_pitch, int tid) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + THREADS * i;
            if (TOTAL % THREADS == 0 || c < TOTAL) {
                const int row = c / CPR, cc = c - row * CPR;
                *reinterpret_cast<uint4*>(lds + row * l_pitch + cc * VEC) =
                    PF ? r[PF ? i : 0] : *reinterpret_cast<const uint4*>(src + row * pitch + cc * VEC);
            }
        }
    }
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// the wave's LDS writes (P tile) must have landed before the same wave reads them as MFMA operands
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int DK, int DV, bool PF>
__global__ __launch_bounds__(256) void nl_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ vt, T* __restrict__ o,
                                                         float* __restrict__ lse, int Nq, int Nk) {
    using M = Mma<T>;
    constexpr int KB = 64, PAD = M::PAD, KSQ = DK / M::KS, KSP = KB / M::KS, NT = DV / 32;
    constexpr int KP = DK + PAD, VP = KB + PAD;
    __shared__ __attribute__((aligned(16))) T smem[KB * KP + DV * VP + 4 * 32 * VP];
    T* Ks = smem;                       // [KB][KP]   keys x dk
    T* Vs = Ks + KB * KP;               // [DV][VP]   dv x keys   (V^T)
    T* Ps = Vs + DV * VP;               // [4][32][VP] per wave: queries x keys
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
    const T* qrow = q + ((long long)b * Nq + q0 + r) * DK;
    const T* kb = k + (long long)b * Nk * DK;
    const T* vtb = vt + (long long)b * DV * Nk;
    T* Pw = Ps + wave * 32 * VP;
    typename M::Frag qf[KSQ];
#pragma unroll
    for (int s = 0; s < KSQ; ++s) qf[s] = M::load(qrow, s, h);

    Stager<T, KB, DK, PF> kst;
    Stager<T, DV, KB, PF> vst;
    // sweep 1: row maxima
    float m = -3.0e38f;
    kst.load(kb, DK, tid);
    for (int k0 = 0; k0 < Nk; k0 += KB) {
        __syncthreads();
        kst.store(Ks, KP, tid);
        __syncthreads();
        kst.load(kb + (long long)(k0 + KB < Nk ? k0 + KB : 0) * DK, DK, tid);     // next block (wraps to sweep 2's first)
#pragma unroll
        for (int t = 0; t < KB / 32; ++t) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KSQ; ++s) acc = M::mma(M::load(Ks + (32 * t + r) * KP, s, h), qf[s], acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, acc[i]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));

    // sweep 2: P = exp(S - m), l = sum P, O^T += V^T P^T
    float l = 0.f;
    f32x16 oacc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) oacc[n] = zero16();
    vst.load(vtb, Nk, tid);
    for (int k0 = 0; k0 < Nk; k0 += KB) {
        __syncthreads();
        kst.store(Ks, KP, tid);
        vst.store(Vs, VP, tid);
        __syncthreads();
        if (k0 + KB < Nk) {
            kst.load(kb + (long long)(k0 + KB) * DK, DK, tid);
            vst.load(vtb + k0 + KB, Nk, tid);
        }
#pragma unroll
        for (int t = 0; t < KB / 32; ++t) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KSQ; ++s) acc = M::mma(M::load(Ks + (32 * t + r) * KP, s, h), qf[s], acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {        // rows (keys) 32t + 8g + 4h + i, column (query) r
                const float p0 = __expf(acc[4 * g] - m), p1 = __expf(acc[4 * g + 1] - m);
                const float p2 = __expf(acc[4 * g + 2] - m), p3 = __expf(acc[4 * g + 3] - m);
                l += (p0 + p1) + (p2 + p3);
                M::store4(Pw + r * VP + 32 * t + 8 * g + 4 * h, p0, p1, p2, p3);
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int s = 0; s < KSP; ++s)
                oacc[n] = M::mma(M::load(Vs + (32 * n + r) * VP, s, h), M::load(Pw + r * VP, s, h), oacc[n]);
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (h == 0) lse[(long long)b * Nq + q0 + r] = m + __logf(l);
    T* orow = o + ((long long)b * Nq + q0 + r) * DV;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)              // rows (dv) 32n + 8g + 4h + i of O^T, column (query) r
            M::store4(orow + 32 * n + 8 * g + 4 * h, oacc[n][4 * g] * inv, oacc[n][4 * g + 1] * inv,
                      oacc[n][4 * g + 2] * inv, oacc[n][4 * g + 3] * inv);
}

// ------------------------------------------------------------------------------------------------ dQ
// dS = P * (dP - delta),  dP = dO V^T,  delta[q] = sum_d dO[q][d] O[q][d] (computed here, and written out for the
// dK / dV kernel that follows on the same stream),  dQ = dS K
template <typename T, int DK, int DV, bool PF>
__global__ __launch_bounds__(256) void nl_attn_bwd_q_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                           const T* __restrict__ kt, const T* __restrict__ v,
                                                           const T* __restrict__ dO, const T* __restrict__ o,
                                                           const float* __restrict__ lse,
                                                           float* __restrict__ delta, T* __restrict__ dq,
                                                           int Nq, int Nk) {
    using M = Mma<T>;
    constexpr int KB = 64, PAD = M::PAD, KSQ = DK / M::KS, KSV = DV / M::KS, KSP = KB / M::KS, MT = (DK + 31) / 32;
    constexpr int KP = DK + PAD, VP = DV + PAD, TP = KB + PAD;
    __shared__ __attribute__((aligned(16))) T smem[KB * KP + KB * VP + 32 * MT * TP + 4 * 32 * TP];
    T* Ks = smem;                       // [KB][KP]        keys x dk
    T* Vs = Ks + KB * KP;               // [KB][VP]        keys x dv
    T* Kts = Vs + KB * VP;              // [32 MT][TP]     dk x keys (K^T), rows >= DK stay zero
    T* Ps = Kts + 32 * MT * TP;         // [4][32][TP]     per wave: queries x keys (dS)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
    const long long qi = (long long)b * Nq + q0 + r;
    const T* kb = k + (long long)b * Nk * DK;
    const T* ktb = kt + (long long)b * DK * Nk;
    const T* vb = v + (long long)b * Nk * DV;
    T* Pw = Ps + wave * 32 * TP;
    for (int c = tid; c < 32 * MT * TP; c += 256) Kts[c] = T(0);
    typename M::Frag qf[KSQ], dof[KSV];
#pragma unroll
    for (int s = 0; s < KSQ; ++s) qf[s] = M::load(q + qi * DK, s, h);
#pragma unroll
    for (int s = 0; s < KSV; ++s) dof[s] = M::load(dO + qi * DV, s, h);
    float Dl = 0.f;
#pragma unroll
    for (int s = 0; s < KSV; ++s) Dl += M::dot(dof[s], M::load(o + qi * DV, s, h));     // this lane's half of the row
    Dl += __shfl_xor(Dl, 32, 64);
    if (h == 0) delta[qi] = Dl;
    const float L = lse[qi];
    f32x16 dqacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dqacc[mt] = zero16();
    Stager<T, KB, DK, PF> kst;
    Stager<T, KB, DV, PF> vst;
    Stager<T, DK, KB, PF> ktst;
    kst.load(kb, DK, tid);
    vst.load(vb, DV, tid);
    ktst.load(ktb, Nk, tid);
    for (int k0 = 0; k0 < Nk; k0 += KB) {
        __syncthreads();
        kst.store(Ks, KP, tid);
        vst.store(Vs, VP, tid);
        ktst.store(Kts, TP, tid);
        __syncthreads();
        if (k0 + KB < Nk) {
            kst.load(kb + (long long)(k0 + KB) * DK, DK, tid);
            vst.load(vb + (long long)(k0 + KB) * DV, DV, tid);
            ktst.load(ktb + k0 + KB, Nk, tid);
        }
#pragma unroll
        for (int t = 0; t < KB / 32; ++t) {
            f32x16 sacc = zero16(), pacc = zero16();
#pragma unroll
            for (int s = 0; s < KSQ; ++s) sacc = M::mma(M::load(Ks + (32 * t + r) * KP, s, h), qf[s], sacc);
#pragma unroll
            for (int s = 0; s < KSV; ++s) pacc = M::mma(M::load(Vs + (32 * t + r) * VP, s, h), dof[s], pacc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float ds[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ds[i] = __expf(sacc[4 * g + i] - L) * (pacc[4 * g + i] - Dl);
                M::store4(Pw + r * TP + 32 * t + 8 * g + 4 * h, ds[0], ds[1], ds[2], ds[3]);
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < KSP; ++s)
                dqacc[mt] = M::mma(M::load(Kts + (32 * mt + r) * TP, s, h), M::load(Pw + r * TP, s, h), dqacc[mt]);
    }
    T* drow = dq + qi * DK;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * mt + 8 * g + 4 * h;
            if (d0 < DK)
                M::store4(drow + d0, dqacc[mt][4 * g], dqacc[mt][4 * g + 1], dqacc[mt][4 * g + 2], dqacc[mt][4 * g + 3]);
        }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// per block of 128 keys: dV = P^T dO, dK = dS^T Q, sweeping the queries.  EIGHT waves: wave (kt, half) belongs to key
// tile kt = wave & 3.  Per block of QB queries, in two stages separated by a barrier:
//   1. the waves of a key tile share its query tiles: S = Q K^T, dP = dO V^T for one 32-query tile each, P^T and dS^T
//      (keys x queries) written to the key tile's LDS scratch;
//   2. they share its OUTPUT: wave `half` accumulates dV^T for half of the dv range and one 32-row tile of dK^T.
// Half the accumulators per wave (<= 168 registers: two waves per SIMD, which is what hides the LDS and staging latency
// here) at the staging traffic of a 128-key block.  The operands of stage 2 whose contraction index (the query) is the slow
// one of their natural layout -- dO^T and Q^T -- are read out of the row-major Q / dO tiles with transposing LDS reads
// (Mma::load_t); the transposed global copies the caller used to make (and the second pair of tiles) are gone.
template <typename T, int DK, int DV, int QB>
__global__ __launch_bounds__(512) void nl_attn_bwd_kv_kernel(const T* __restrict__ q, const T* __restrict__ qt,
                                                            const T* __restrict__ k, const T* __restrict__ v,
                                                            const T* __restrict__ dO, const T* __restrict__ dOt,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta, T* __restrict__ dk,
                                                            T* __restrict__ dv, float* __restrict__ part, int Nq,
                                                            int Nk) {
    using M = Mma<T>;
    constexpr int PAD = M::PAD, KSQ = DK / M::KS, KSV = DV / M::KS, KSP = QB / M::KS, NT = DV / 32, MT = (DK + 31) / 32;
    constexpr int NH = NT / 2;                 // dV^T tiles per wave
    static_assert(NT % 2 == 0 && MT <= 2, "the two waves of a key tile split the dv range and the dk tiles");
    constexpr int QP = DK + PAD, OP = DV + PAD, TP = QB + PAD;
    constexpr int STAT_T = (2 * QB * (int)sizeof(float)) / (int)sizeof(T);     // Ls + Ds in units of T
    // (Qs rows are read up to column 32 MT - 1 by the transposing reads: columns >= DK belong to output rows that are never
    //  stored; the slack behind the last row keeps those reads inside the array)
    __shared__ __attribute__((aligned(16))) T smem[QB * QP + 64 + QB * OP + STAT_T + 2 * 4 * 32 * TP];
    T* Qs = smem;                       // [QB][QP]      queries x dk
    T* dOs = Qs + QB * QP + 64;         // [QB][OP]      queries x dv
    float* Ls = reinterpret_cast<float*>(dOs + QB * OP);
    float* Ds = Ls + QB;
    T* Pts = reinterpret_cast<T*>(Ds + QB);    // [4][32][TP]  per key tile: keys x queries (P^T)
    T* dSts = Pts + 4 * 32 * TP;               // [4][32][TP]  per key tile: keys x queries (dS^T)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int kt = wave & 3, half = wave >> 2;
    const int b = blockIdx.y, key0 = blockIdx.x * 128 + kt * 32;
    const long long ki = (long long)b * Nk + key0 + r;
    const T* qb = q + (long long)b * Nq * DK;
    const T* dob = dO + (long long)b * Nq * DV;
    T* Ptw = Pts + kt * 32 * TP;
    T* dStw = dSts + kt * 32 * TP;
    typename M::Frag kf[KSQ], vf[KSV];
#pragma unroll
    for (int s = 0; s < KSQ; ++s) kf[s] = M::load(k + ki * DK, s, h);
#pragma unroll
    for (int s = 0; s < KSV; ++s) vf[s] = M::load(v + ki * DV, s, h);
    f32x16 dvacc[NH], dkacc = zero16();
#pragma unroll
    for (int n = 0; n < NH; ++n) dvacc[n] = zero16();
    // gridDim.z workgroups share a key block, each sweeping its own range of the queries (more workgroups than CUs even
    // when B * Nk / 128 is small); with more than one of them the results are fp32 partial sums in `part`
    const int q_per = Nq / gridDim.z, q_lo = blockIdx.z * q_per, q_hi = q_lo + q_per;
    for (int q0 = q_lo; q0 < q_hi; q0 += QB) {
        __syncthreads();
        Stager<T, QB, DK, false, 512> qst;
        Stager<T, QB, DV, false, 512> ost;
        qst.load(qb + (long long)q0 * DK, DK, tid);   qst.store(Qs, QP, tid);
        ost.load(dob + (long long)q0 * DV, DV, tid);  ost.store(dOs, OP, tid);
        if (tid < QB) Ls[tid] = lse[(long long)b * Nq + q0 + tid];
        else if (tid < 2 * QB) Ds[tid - QB] = delta[(long long)b * Nq + q0 + tid - QB];
        __syncthreads();
        // stage 1: query tiles half, half + 2, ... of the block
#pragma unroll
        for (int t = 0; t < QB / 32; ++t) {
            if ((t & 1) != half) continue;
            f32x16 sacc = zero16(), pacc = zero16();      // rows: queries, column: key r
#pragma unroll
            for (int s = 0; s < KSQ; ++s) sacc = M::mma(M::load(Qs + (32 * t + r) * QP, s, h), kf[s], sacc);
#pragma unroll
            for (int s = 0; s < KSV; ++s) pacc = M::mma(M::load(dOs + (32 * t + r) * OP, s, h), vf[s], pacc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qq = 32 * t + 8 * g + 4 * h;
                const float4 Lv = *reinterpret_cast<const float4*>(Ls + qq);
                const float4 Dv = *reinterpret_cast<const float4*>(Ds + qq);
                const float p0 = __expf(sacc[4 * g] - Lv.x), p1 = __expf(sacc[4 * g + 1] - Lv.y);
                const float p2 = __expf(sacc[4 * g + 2] - Lv.z), p3 = __expf(sacc[4 * g + 3] - Lv.w);
                M::store4(Ptw + r * TP + qq, p0, p1, p2, p3);
                M::store4(dStw + r * TP + qq, p0 * (pacc[4 * g] - Dv.x), p1 * (pacc[4 * g + 1] - Dv.y),
                          p2 * (pacc[4 * g + 2] - Dv.z), p3 * (pacc[4 * g + 3] - Dv.w));
            }
        }
        __syncthreads();
        // stage 2: this wave's share of the key tile's outputs, over all QB queries
#pragma unroll
        for (int n = 0; n < NH; ++n)
#pragma unroll
            for (int s = 0; s < KSP; ++s)
                dvacc[n] = M::mma(M::load_t(dOs, OP, 32 * (half * NH + n), s, lane), M::load(Ptw + r * TP, s, h), dvacc[n]);
        if (half < MT) {
#pragma unroll
            for (int s = 0; s < KSP; ++s)
                dkacc = M::mma(M::load_t(Qs, QP, 32 * half, s, lane), M::load(dStw + r * TP, s, h), dkacc);
        }
    }
    if (gridDim.z > 1) {               // fp32 partial sums: [z][B * Nk][DV + DK]
        float* prow = part + ((long long)blockIdx.z * gridDim.y * Nk + ki) * (DV + DK);
#pragma unroll
        for (int n = 0; n < NH; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(prow + 32 * (half * NH + n) + 8 * g + 4 * h) =
                    make_float4(dvacc[n][4 * g], dvacc[n][4 * g + 1], dvacc[n][4 * g + 2], dvacc[n][4 * g + 3]);
        if (half < MT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * half + 8 * g + 4 * h;
                if (d0 < DK)
                    *reinterpret_cast<float4*>(prow + DV + d0) =
                        make_float4(dkacc[4 * g], dkacc[4 * g + 1], dkacc[4 * g + 2], dkacc[4 * g + 3]);
            }
        }
        return;
    }
    T* dvrow = dv + ki * DV;
#pragma unroll
    for (int n = 0; n < NH; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            M::store4(dvrow + 32 * (half * NH + n) + 8 * g + 4 * h, dvacc[n][4 * g], dvacc[n][4 * g + 1],
                      dvacc[n][4 * g + 2], dvacc[n][4 * g + 3]);
    if (half < MT) {
        T* dkrow = dk + ki * DK;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * half + 8 * g + 4 * h;
            if (d0 < DK) M::store4(dkrow + d0, dkacc[4 * g], dkacc[4 * g + 1], dkacc[4 * g + 2], dkacc[4 * g + 3]);
        }
    }
}

// dk / dv (storage type) = sum over the query splits of the fp32 partial rows [z][rows][DV + DK]
template <typename T>
__global__ void nl_attn_reduce_kernel(const float* __restrict__ part, T* __restrict__ dk, T* __restrict__ dv,
                                      long long rows, int DK, int DV, int nsplit) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one 4-element group
    const int gpr = (DV + DK) / 4;
    if (i >= rows * gpr) return;
    const long long row = i / gpr;
    const int c = (int)(i - row * gpr) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {
        const float4 p = *reinterpret_cast<const float4*>(part + ((long long)z * rows + row) * (DV + DK) + c);
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    T* out = c < DV ? dv + row * DV + c : dk + row * DK + (c - DV);
    Mma<T>::store4(out, acc.x, acc.y, acc.z, acc.w);
}

// MSG_ATTN_PREFETCH (A/B switch): bit 0 forward, bit 1 dQ kernel stage through registers one block ahead.  Default 0: measured slower in every kernel (the registers it holds cost more than the latency it hides).
int prefetch_mode() {
    static const int mode = msg_tunable("MSG_ATTN_PREFETCH", 0);
    return mode;
}

bool shapes_ok(int B, int Nq, int Nk, int dk, int dv) {
    return B > 0 && Nq > 0 && Nk > 0 && Nq % 128 == 0 && Nk % 128 == 0 && ((dk == 48 && dv == 192) || (dk == 16 && dv == 64));
}

template <typename T, int DK, int DV>
int launch_fwd(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int Nq, int Nk, hipStream_t s) {
    if (prefetch_mode() & 1)
        hipLaunchKernelGGL((nl_attn_fwd_kernel<T, DK, DV, true>), dim3(Nq / 128, B), dim3(256), 0, s, (const T*)q,
                           (const T*)k, (const T*)vt, (T*)o, lse, Nq, Nk);
    else
        hipLaunchKernelGGL((nl_attn_fwd_kernel<T, DK, DV, false>), dim3(Nq / 128, B), dim3(256), 0, s, (const T*)q,
                           (const T*)k, (const T*)vt, (T*)o, lse, Nq, Nk);
    return MSG_CHECK_LAUNCH();
}

template <typename T, int DK, int DV>
int launch_bwd(const void* q, const void* qt, const void* k, const void* kt, const void* v, const void* dO,
               const void* dOt, const void* o, const float* lse, float* delta, void* dq, void* dk, void* dv,
               float* part, int nsplit, int B, int Nq, int Nk, hipStream_t s) {
    constexpr int QB = sizeof(T) == 2 ? 64 : 32;
    if (prefetch_mode() & 2)
        hipLaunchKernelGGL((nl_attn_bwd_q_kernel<T, DK, DV, true>), dim3(Nq / 128, B), dim3(256), 0, s, (const T*)q,
                           (const T*)k, (const T*)kt, (const T*)v, (const T*)dO, (const T*)o, lse, delta, (T*)dq, Nq, Nk);
    else
        hipLaunchKernelGGL((nl_attn_bwd_q_kernel<T, DK, DV, false>), dim3(Nq / 128, B), dim3(256), 0, s, (const T*)q,
                           (const T*)k, (const T*)kt, (const T*)v, (const T*)dO, (const T*)o, lse, delta, (T*)dq, Nq, Nk);
    if (hipGetLastError() != hipSuccess) return MSG_ELAUNCH;
    hipLaunchKernelGGL((nl_attn_bwd_kv_kernel<T, DK, DV, QB>), dim3(Nk / 128, B, nsplit), dim3(512), 0, s, (const T*)q,
                       (const T*)qt, (const T*)k, (const T*)v, (const T*)dO, (const T*)dOt, lse, delta, (T*)dk, (T*)dv,
                       part, Nq, Nk);
    if (nsplit > 1) {
        if (hipGetLastError() != hipSuccess) return MSG_ELAUNCH;
        const long long rows = (long long)B * Nk, groups = rows * ((DK + DV) / 4);
        hipLaunchKernelGGL((nl_attn_reduce_kernel<T>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, part,
                           (T*)dk, (T*)dv, rows, DK, DV, nsplit);
    }
    return MSG_CHECK_LAUNCH();
}

}  // namespace

extern "C" int msg_nonlocal_attention_supported(int B, int Nq, int Nk, int dk, int dv) {
    return shapes_ok(B, Nq, Nk, dk, dv) ? 1 : 0;
}

extern "C" int msg_nonlocal_attention_fwd(const void* q, const void* k, const void* vt, void* o, float* lse, int dtype,
                                          int B, int Nq, int Nk, int dk, int dv, void* stream) {
    if (!q || !k || !vt || !o || !lse) return MSG_EINVAL;
    if (!shapes_ok(B, Nq, Nk, dk, dv)) return MSG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        return dk == 48 ? launch_fwd<bf16_t, 48, 192>(q, k, vt, o, lse, B, Nq, Nk, s)
                        : launch_fwd<bf16_t, 16, 64>(q, k, vt, o, lse, B, Nq, Nk, s);
    if (dtype == MSG_F32)
        return dk == 48 ? launch_fwd<float, 48, 192>(q, k, vt, o, lse, B, Nq, Nk, s)
                        : launch_fwd<float, 16, 64>(q, k, vt, o, lse, B, Nq, Nk, s);
    return MSG_EUNSUPPORTED;
}

// How many ways msg_nonlocal_attention_bwd splits the query sweep of its dK / dV kernel (1, 2, 4 or 8: enough
// workgroups for the 256 CUs); with more than one the caller provides `workspace`, nsplit * B * Nk * (dk + dv) floats.
extern "C" int msg_nonlocal_attention_bwd_splits(int B, int Nq, int Nk) {
    static const int forced = msg_tunable("MSG_ATTN_SPLITS", 0);
    if (forced > 0 && Nq % (forced * 128) == 0) return forced;
    int nsplit = 1;
    while (nsplit < 8 && (long long)B * (Nk / 128) * nsplit < 384 && (Nq / nsplit) % 128 == 0 && Nq / nsplit >= 512)
        nsplit *= 2;
    return nsplit;
}

extern "C" int msg_nonlocal_attention_bwd(const void* q, const void* qt, const void* k, const void* kt, const void* v,
                                          const void* dO, const void* dOt, const void* o, const float* lse,
                                          float* delta, void* dq, void* dk_out, void* dv_out, float* workspace,
                                          int dtype, int B, int Nq, int Nk, int dk, int dv, void* stream) {
    if (!q || !k || !kt || !v || !dO || !o || !lse || !delta || !dq || !dk_out || !dv_out) return MSG_EINVAL;   // (qt, dOt: no longer read)
    if (!shapes_ok(B, Nq, Nk, dk, dv)) return MSG_EUNSUPPORTED;
    const int nsplit = msg_nonlocal_attention_bwd_splits(B, Nq, Nk);
    if (nsplit > 1 && !workspace) return MSG_EINVAL;
    float* part = workspace;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        return dk == 48 ? launch_bwd<bf16_t, 48, 192>(q, qt, k, kt, v, dO, dOt, o, lse, delta, dq, dk_out, dv_out, part, nsplit, B, Nq, Nk, s)
                        : launch_bwd<bf16_t, 16, 64>(q, qt, k, kt, v, dO, dOt, o, lse, delta, dq, dk_out, dv_out, part, nsplit, B, Nq, Nk, s);
    if (dtype == MSG_F32)
        return dk == 48 ? launch_bwd<float, 48, 192>(q, qt, k, kt, v, dO, dOt, o, lse, delta, dq, dk_out, dv_out, part, nsplit, B, Nq, Nk, s)
                        : launch_bwd<float, 16, 64>(q, qt, k, kt, v, dO, dOt, o, lse, delta, dq, dk_out, dv_out, part, nsplit, B, Nq, Nk, s);
    return MSG_EUNSUPPORTED;
}
