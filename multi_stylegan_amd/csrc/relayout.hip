// Weight re-layout for the contraction kernels: ONE pass over a parameter produces every K-contiguous image the
// forward / data-gradient / modulation kernels read from it.
//
//   w     [O][I][T]  fp32, the parameter as the reference stores it (T = kh*kw taps, tap fastest)
//   fwd   [O][T][Ck] (or [T][O][Ck] when t_major: the 2x2 transposed conv's 4*O output rows)   = gain * w, zero for i >= I
//   dgrad [I][T][Ok] with the taps flipped (or not: transposed conv)                            = gain * w, zero for o >= O
//   wsq   [O][I]     fp32 sum over taps of w^2 (demodulation)
//
// Every output is optional.  Before this kernel each image was a torch transpose-copy (plus a zero fill, plus a scale),
// ~330 small launches and ~5 ms per training step, re-done after every optimiser step because the weights changed.
// A workgroup moves a 32 (o) x 32 (i) x T tile through LDS: the read is contiguous along (i, t), the forward image is
// written contiguous along i and the gradient image contiguous along o.
#include "msg_common.h"

constexpr int RL_T = 16;                      // max taps held per tile (kernels up to 4x4)

template <typename TO, int TC>       // TC: compile-time tap count (1, 4, 9, 16: index divisions become shifts / multiplies), 0 = runtime
__global__ __launch_bounds__(256) void relayout_weight_kernel(const float* __restrict__ w, TO* __restrict__ fwd,
                                                              TO* __restrict__ dgr, float* __restrict__ wsq,
                                                              int O, int I, int Trt, int Ck, int Ok, int flip, int t_major,
                                                              float gain) {
    const int T = TC ? TC : Trt;
    __shared__ float tile[32][32 * RL_T + 1];
    const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tid = threadIdx.x;
    const int row_len = 32 * T;                                   // floats of one o row of the tile (i-major, tap fastest)
    // ---- load (zero outside the parameter: those cells become the padding of the images)
    for (int e = tid; e < 32 * row_len; e += 256) {
        const int ro = e / row_len, c = e - ro * row_len;         // c = il * T + t
        const int o = o0 + ro, il = c / T;
        const int i = i0 + il;
        tile[ro][c] = (o < O && i < I) ? w[((size_t)o * I + i) * T + (c - il * T)] : 0.f;
    }
    __syncthreads();
    // ---- forward image: lanes along i
    if (fwd) {
        for (int e = tid; e < 32 * T * 32; e += 256) {
            const int il = e & 31, rt = e >> 5;                   // rt = ro * T + t
            const int ro = rt / T, t = rt - ro * T;
            const int o = o0 + ro, i = i0 + il;
            if (o < O && i < Ck) {
                const size_t row = t_major ? (size_t)t * O + o : (size_t)o * T + t;
                store_from_f32(fwd + row * Ck + i, tile[ro][il * T + t] * gain);
            }
        }
    }
    // ---- data-gradient image: lanes along o
    if (dgr) {
        for (int e = tid; e < 32 * T * 32; e += 256) {
            const int ro = e & 31, it = e >> 5;                   // it = il * T + t
            const int il = it / T, t = it - il * T;
            const int o = o0 + ro, i = i0 + il;
            if (i < I && o < Ok) {
                const int td = flip ? T - 1 - t : t;
                store_from_f32(dgr + ((size_t)i * T + td) * Ok + o, tile[ro][il * T + t] * gain);
            }
        }
    }
    if (wsq) {
        for (int e = tid; e < 32 * 32; e += 256) {
            const int il = e & 31, ro = e >> 5;
            const int o = o0 + ro, i = i0 + il;
            if (o < O && i < I) {
                float s = 0.f;
                for (int t = 0; t < T; ++t) { const float v = tile[ro][il * T + t]; s = fmaf(v, v, s); }
                wsq[(size_t)o * I + i] = s;
            }
        }
    }
}

extern "C" int msg_relayout_weight(const float* w, void* fwd, void* dgrad, float* wsq, int dtype,
                                   int O, int I, int T, int Ck, int Ok, int flip, int t_major, float gain,
                                   void* stream) {
    if (!w || O <= 0 || I <= 0 || T <= 0 || (fwd && Ck < I) || (dgrad && Ok < O)) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    if (T > RL_T) return MSG_EUNSUPPORTED;
    // the tile grid must also cover the zero padding of the images (i up to Ck, o up to Ok)
    const int o_ext = dgrad && Ok > O ? Ok : O, i_ext = fwd && Ck > I ? Ck : I;
    dim3 grid((o_ext + 31) / 32, (i_ext + 31) / 32);
    if (grid.y > 65535) return MSG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define RL_LAUNCH(TC_)                                                                                                 \
    do {                                                                                                               \
        if (dtype == MSG_BF16)                                                                                         \
            hipLaunchKernelGGL((relayout_weight_kernel<bf16_t, TC_>), grid, dim3(256), 0, s, w, (bf16_t*)fwd,          \
                               (bf16_t*)dgrad, wsq, O, I, T, Ck, Ok, flip, t_major, gain);                             \
        else                                                                                                           \
            hipLaunchKernelGGL((relayout_weight_kernel<float, TC_>), grid, dim3(256), 0, s, w, (float*)fwd,            \
                               (float*)dgrad, wsq, O, I, T, Ck, Ok, flip, t_major, gain);                              \
    } while (0)
    switch (T) {
        case 1: RL_LAUNCH(1); break;
        case 4: RL_LAUNCH(4); break;
        case 9: RL_LAUNCH(9); break;
        case 16: RL_LAUNCH(16); break;
        default: RL_LAUNCH(0); break;
    }
#undef RL_LAUNCH
    return MSG_CHECK_LAUNCH();
}
