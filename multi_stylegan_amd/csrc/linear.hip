// Equalized-lr fully connected layers with FEW rows (the 8-layer mapping network, the per-layer style affines, the
// discriminator's classification head: M = batch = 16..64 rows, N, K <= ~1024, fp32).
//
// These are latency-bound, not throughput-bound: 1 MiB of weights and a few MFLOP per call, ~140 calls per training
// step.  Pushing them through the 128x128 MFMA conv tile costs ~40 us per call (4 workgroups walk K serially) plus
// layout copies; here every call is ONE launch that spreads the weight matrix over the whole chip and reads it
// exactly once, fully coalesced, with the batch rows kept in registers / served from L1:
//
//   fprop  y[m][n]  = gain * sum_k x[m][k] * W[n][k] + bias_gain * bias[n]     one wave per output column n
//   dgrad  gx[m][k] = gain * sum_n gy[m][n] * W[n][k]                          16 waves per 64 columns k, rows n split
//   wgrad  gW[n][k] = gain * sum_m gy[m][n] * x[m][k],  gb[n] = bias_gain * sum_m gy[m][n]   one workgroup per n
//
// The three are closed under differentiation (each one's derivatives are the other two), which is what the R1 and
// path-length regularisers need from the style affines and the classification head.
// Reference: multi_stylegan/equalized_layer.py (EqualizedLinear.forward: F.linear(x, W * scale, bias * scale_b)).
#include "msg_common.h"

constexpr int LIN_MB = 16;         // batch rows per accumulator block

constexpr int LIN_KC = 1024;       // K elements whose weights one wave keeps in registers (16 per lane)

// (Measured alternative: 4 output columns per wave to cut the L1/L2 re-reads of x by 4 -- 2x SLOWER at M=16, N=K=512:
//  the kernel is bound by the latency chain of one wave, not by traffic, so more, lighter waves win.)
// (Round 4: 16-byte loads -- a lane takes four consecutive k -- cut a launch from 10.9 to 7.9 us, but regroup the fp32 sum, and
//  the path-length regulariser's second-order gradients, which amplify the mapping network's last bits, moved from 9e-4 to
//  3.7e-3 of the oracle's on one parameter (gate 2e-3, tests/test_hip_models.py).  Not kept: 0.07 ms per iteration.)
__device__ __forceinline__ void linear_fprop_body(const float* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                  int M, int N, int K, float gain, float bias_gain) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (n >= N) return;
    const float* wn = w + (size_t)n * K;
    const float bv = bias ? bias_gain * bias[n] : 0.f;
    for (int m0 = 0; m0 < M; m0 += LIN_MB) {
        float acc[LIN_MB];
#pragma unroll
        for (int mm = 0; mm < LIN_MB; ++mm) acc[mm] = 0.f;
        for (int kc = 0; kc < K; kc += LIN_KC) {
            float wr[LIN_KC / 64];
#pragma unroll
            for (int j = 0; j < LIN_KC / 64; ++j) {
                const int k = kc + lane + 64 * j;
                wr[j] = k < K ? wn[k] : 0.f;
            }
#pragma unroll
            for (int mm = 0; mm < LIN_MB; ++mm) {
                const int m = m0 + mm;
                if (m >= M) break;
                const float* xm = x + (size_t)m * ldx;
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < LIN_KC / 64; ++j) {
                    const int k = kc + lane + 64 * j;
                    a = fmaf(wr[j], k < K ? xm[k] : 0.f, a);
                }
                acc[mm] += a;
            }
        }
#pragma unroll
        for (int mm = 0; mm < LIN_MB; ++mm) {
            const int m = m0 + mm;
            if (m >= M) break;
            const float tot = wave_sum(acc[mm]);
            if (lane == 0) y[(size_t)m * N + n] = gain * tot + bv;
        }
    }
}

__global__ __launch_bounds__(256) void linear_fprop_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int M, int N, int K, float gain, float bias_gain) {
    linear_fprop_body(x, K, w, bias, y, M, N, K, gain, bias_gain);
}

// GROUPED variants (blockIdx.y = group): G independent layers of the same shape in one launch -- the generator's 20 style
// affines all read the latent [M][L][K] (group g reads slot slot[g], rows L*K apart) and are known before the first
// convolution runs, so they cost one launch forward and two backward instead of 20 + 40 (each ~12 us of an otherwise
// idle chip).  w / bias come from pointer tables (the layers keep their own parameters); outputs are [G][...] stacks.
__global__ __launch_bounds__(256) void linear_grouped_fprop_kernel(const float* __restrict__ x, const int* __restrict__ slot,
                                                                   const float* const* __restrict__ w, const float* const* __restrict__ bias,
                                                                   float* __restrict__ y, int M, int N, int K, int L,
                                                                   float gain, float bias_gain) {
    const int g = blockIdx.y;
    linear_fprop_body(x + (size_t)slot[g] * K, (long long)L * K, w[g], bias ? bias[g] : nullptr, y + (size_t)g * M * N,
                      M, N, K, gain, bias_gain);
}

// One workgroup of 16 waves per 64 output columns k.  The batch rows' gy values are staged TRANSPOSED in LDS
// ([n][16 rows], so one ds_read_b128 broadcast hands a wave 4 rows' factors); wave q walks its 1/16 of the rows n of W
// (each row segment is one coalesced 256-B load) with 16 batch rows in registers, then the 16 partial sums meet in the
// same LDS where thread (m, k) adds them in a fixed order (deterministic, no atomics).
constexpr int LIN_NCH = 1024;      // rows of W per staged chunk: 1024 x 16 floats = 64 KiB of LDS
__device__ __forceinline__ void linear_dgrad_body(float* __restrict__ sh, const float* __restrict__ gy, const float* __restrict__ w,
                                                  float* __restrict__ gx, int M, int N, int K, float gain) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * 64 + lane;
    const bool kok = k < K;
    const float* wk = w + (kok ? k : 0);
    for (int m0 = 0; m0 < M; m0 += LIN_MB) {
        float acc[LIN_MB];
#pragma unroll
        for (int mm = 0; mm < LIN_MB; ++mm) acc[mm] = 0.f;
        for (int nc = 0; nc < N; nc += LIN_NCH) {
            const int nn = min(LIN_NCH, N - nc);
            __syncthreads();
            for (int e = threadIdx.x; e < LIN_MB * nn; e += 1024) {       // coalesced along n, transposed into LDS
                const int mm = e / nn, n = e - mm * nn;
                sh[n * LIN_MB + mm] = (m0 + mm < M) ? gy[(size_t)(m0 + mm) * N + nc + n] : 0.f;
            }
            __syncthreads();
            const int nper = (nn + 15) / 16;
            const int na = q * nper, nb = min(nn, na + nper);
#pragma unroll 4
            for (int n = na; n < nb; ++n) {
                const float wv = wk[(size_t)(nc + n) * K];
                const f32x4* g4 = reinterpret_cast<const f32x4*>(sh + n * LIN_MB);
#pragma unroll
                for (int v = 0; v < LIN_MB / 4; ++v) {
                    const f32x4 g = g4[v];
                    acc[4 * v + 0] = fmaf(g[0], wv, acc[4 * v + 0]);
                    acc[4 * v + 1] = fmaf(g[1], wv, acc[4 * v + 1]);
                    acc[4 * v + 2] = fmaf(g[2], wv, acc[4 * v + 2]);
                    acc[4 * v + 3] = fmaf(g[3], wv, acc[4 * v + 3]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < LIN_MB; ++mm) sh[(q * LIN_MB + mm) * 64 + lane] = acc[mm];
        __syncthreads();
        {
            const int mm = threadIdx.x >> 6;        // 16 waves <-> 16 batch rows
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += sh[(r * LIN_MB + mm) * 64 + lane];
            if (m0 + mm < M && kok) gx[(size_t)(m0 + mm) * K + k] = gain * s;
        }
    }
}

__global__ __launch_bounds__(1024) void linear_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                            float* __restrict__ gx, int M, int N, int K, float gain) {
    __shared__ __attribute__((aligned(16))) float sh[LIN_NCH * LIN_MB];      // gyT[n][m], later red[q][m][lane]
    linear_dgrad_body(sh, gy, w, gx, M, N, K, gain);
}

__global__ __launch_bounds__(1024) void linear_grouped_dgrad_kernel(const float* __restrict__ gy, const float* const* __restrict__ w,
                                                                    float* __restrict__ gx, int M, int N, int K, float gain) {
    __shared__ __attribute__((aligned(16))) float sh[LIN_NCH * LIN_MB];
    const int g = blockIdx.y;
    linear_dgrad_body(sh, gy + (size_t)g * M * N, w[g], gx + (size_t)g * M * K, M, N, K, gain);
}

__device__ __forceinline__ void linear_wgrad_body(const float* __restrict__ gy, const float* __restrict__ x, long long ldx,
                                                  float* __restrict__ gw, float* __restrict__ gb,
                                                  int M, int N, int K, float gain, float bias_gain) {
    const int n = blockIdx.x;
    const float* gn = gy + n;                       // gy[m][n] = gn[m * N]: block-uniform
    for (int k = threadIdx.x; k < K; k += 256) {
        float a = 0.f;
        for (int m = 0; m < M; ++m) a = fmaf(gn[(size_t)m * N], x[(size_t)m * ldx + k], a);
        gw[(size_t)n * K + k] = gain * a;
    }
    if (gb && threadIdx.x == 0) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += gn[(size_t)m * N];
        gb[n] = bias_gain * s;
    }
}

__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                           float* __restrict__ gw, float* __restrict__ gb,
                                                           int M, int N, int K, float gain, float bias_gain) {
    linear_wgrad_body(gy, x, K, gw, gb, M, N, K, gain, bias_gain);
}

__global__ __launch_bounds__(256) void linear_grouped_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                   const int* __restrict__ slot, float* __restrict__ gw,
                                                                   float* __restrict__ gb, int M, int N, int K, int L,
                                                                   float gain, float bias_gain) {
    const int g = blockIdx.y;
    linear_wgrad_body(gy + (size_t)g * M * N, x + (size_t)slot[g] * K, (long long)L * K, gw + (size_t)g * N * K,
                      gb ? gb + (size_t)g * N : nullptr, M, N, K, gain, bias_gain);
}

// ... with one destination pointer per layer (the layers' slices of a flat gradient store) instead of stacked results
__global__ __launch_bounds__(256) void linear_grouped_wgrad_ptrs_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                        const int* __restrict__ slot, float* const* __restrict__ gw,
                                                                        float* const* __restrict__ gb, int M, int N, int K, int L,
                                                                        float gain, float bias_gain) {
    const int g = blockIdx.y;
    linear_wgrad_body(gy + (size_t)g * M * N, x + (size_t)slot[g] * K, (long long)L * K, gw[g], gb ? gb[g] : nullptr, M, N, K,
                      gain, bias_gain);
}

static bool lin_bad(const void* a, const void* b, const void* c, int M, int N, int K) {
    return !a || !b || !c || M < 0 || N <= 0 || K <= 0;
}

extern "C" int msg_linear_fprop(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                                float gain, float bias_gain, void* stream) {
    if (M == 0) return MSG_OK;
    if (lin_bad(x, w, y, M, N, K)) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_fprop_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K,
                       gain, bias_gain);
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_linear_dgrad(const float* gy, const float* w, float* gx, int M, int N, int K, float gain,
                                void* stream) {
    if (M == 0) return MSG_OK;
    if (lin_bad(gy, w, gx, M, N, K)) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_dgrad_kernel, dim3((K + 63) / 64), dim3(1024), 0, (hipStream_t)stream, gy, w, gx,
                       M, N, K, gain);
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_linear_wgrad(const float* gy, const float* x, float* gw, float* gb, int M, int N, int K,
                                float gain, float bias_gain, void* stream) {
    if (M == 0 && gw && N > 0 && K > 0) {                 // empty batch: the gradient is zero and must still be written
        hipError_t e = hipMemsetAsync(gw, 0, (size_t)N * K * sizeof(float), (hipStream_t)stream);
        if (gb && e == hipSuccess) e = hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), (hipStream_t)stream);
        return e == hipSuccess ? MSG_OK : MSG_ELAUNCH;
    }
    if (lin_bad(gy, x, gw, M, N, K)) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, gy, x, gw, gb, M, N, K, gain,
                       bias_gain);
    return MSG_CHECK_LAUNCH();
}

// ---- grouped entry points (see linear_grouped_fprop_kernel)
extern "C" int msg_linear_grouped_fprop(const float* x, const int* slot, const float* const* w, const float* const* bias,
                                        float* y, int G, int M, int N, int K, int L, float gain, float bias_gain,
                                        void* stream) {
    if (M == 0 || G == 0) return MSG_OK;
    if (lin_bad(x, w, y, M, N, K) || !slot || G < 0 || L <= 0 || G > 65535) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_grouped_fprop_kernel, dim3((N + 3) / 4, G), dim3(256), 0, (hipStream_t)stream, x, slot, w,
                       bias, y, M, N, K, L, gain, bias_gain);
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_linear_grouped_dgrad(const float* gy, const float* const* w, float* gx, int G, int M, int N, int K,
                                        float gain, void* stream) {
    if (M == 0 || G == 0) return MSG_OK;
    if (lin_bad(gy, w, gx, M, N, K) || G < 0 || G > 65535) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_grouped_dgrad_kernel, dim3((K + 63) / 64, G), dim3(1024), 0, (hipStream_t)stream, gy, w, gx,
                       M, N, K, gain);
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_linear_grouped_wgrad(const float* gy, const float* x, const int* slot, float* gw, float* gb, int G,
                                        int M, int N, int K, int L, float gain, float bias_gain, void* stream) {
    if (G == 0) return MSG_OK;
    if (M <= 0 || lin_bad(gy, x, gw, M, N, K) || !slot || G < 0 || L <= 0 || G > 65535) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_grouped_wgrad_kernel, dim3(N, G), dim3(256), 0, (hipStream_t)stream, gy, x, slot, gw, gb,
                       M, N, K, L, gain, bias_gain);
    return MSG_CHECK_LAUNCH();
}

// msg_linear_grouped_wgrad writing layer g's results to gw[g] ([N][K]) and gb[g] ([N]; gb may be NULL): device arrays of G
// device pointers -- the parameters' own slices of a flat gradient store, so that no per-layer copy follows.
extern "C" int msg_linear_grouped_wgrad_ptrs(const float* gy, const float* x, const int* slot, float* const* gw,
                                             float* const* gb, int G, int M, int N, int K, int L, float gain, float bias_gain,
                                             void* stream) {
    if (G == 0) return MSG_OK;
    if (M <= 0 || lin_bad(gy, x, gw, M, N, K) || !slot || G < 0 || L <= 0 || G > 65535) return MSG_EINVAL;
    hipLaunchKernelGGL(linear_grouped_wgrad_ptrs_kernel, dim3(N, G), dim3(256), 0, (hipStream_t)stream, gy, x, slot, gw, gb,
                       M, N, K, L, gain, bias_gain);
    return MSG_CHECK_LAUNCH();
}
