// a2: fused (noise +) bias + leaky-ReLU for gfx950, forward / first derivative / reductions.
// Replaces multi_stylegan/op_static/fused_bias_act_kernel.cu:18-99 and folds the NoiseInjection add
// (multi_stylegan_generator.py:288-292) and the PyTorch-side grad_bias sum (op_static/fused_act.py:35-40)
// into the same pass.  HBM-bound: one 16-byte access per lane per tensor, fp32 arithmetic.
#include "msg_common.h"

struct BiasActParams {
    long long size_x;
    int step_b, size_b, noise_batch, pix, act, grad;
    float alpha, scale;
};

__device__ __forceinline__ float act_apply(float v, float gate, int act, float alpha, float scale) {
    // fused_bias_act_kernel.cu:36-47: act*10+grad in {30,31}: slope chosen by sign of x (fwd) or of ref (grad)
    if (act == 3) v = (gate > 0.f) ? v : v * alpha;
    return v * scale;
}

// ---- generic scalar path (any step_b / size_b) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bias_act_scalar_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                              const T* __restrict__ ref, T* __restrict__ y,
                                                              const float* __restrict__ noise,
                                                              const float* __restrict__ noise_w, BiasActParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.size_x) return;
    float v = load_as_f32(x + i);
    if (p.grad == 2) { store_from_f32(y + i, 0.f); return; }
    const long long q = i / p.step_b;
    if (noise) {
        const long long b = q / p.size_b;
        long long pixel = (p.step_b == 1) ? q / p.size_b : b * p.pix + (i - q * p.step_b);
        if (p.noise_batch == 1) pixel %= p.pix;
        v = fmaf(noise_w[0], noise[pixel], v);
    }
    if (bias) v += bias[q % p.size_b];
    const float gate = (p.grad == 1 && ref) ? load_as_f32(ref + i) : v;
    store_from_f32(y + i, act_apply(v, gate, p.act, p.alpha, p.scale));
}

// ---- float64 (MSG_F64: the `double` of AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79) -------------
// x, bias, ref, y all float64, arithmetic in double exactly as fused_bias_act_kernel.cu:26-47 (x + b, slope by the sign
// of x + b or of ref, * scale; alpha and scale are the float arguments widened).  No noise (our extension is fp32).
__global__ __launch_bounds__(256) void bias_act_f64_kernel(const double* __restrict__ x, const double* __restrict__ bias,
                                                           const double* __restrict__ ref, double* __restrict__ y,
                                                           BiasActParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.size_x) return;
    if (p.grad == 2) { y[i] = 0.0; return; }
    double v = x[i];
    if (bias) v += bias[(i / p.step_b) % p.size_b];
    const double gate = (p.grad == 1 && ref) ? ref[i] : v;
    if (p.act == 3) v = (gate > 0.0) ? v : v * (double)p.alpha;
    y[i] = v * (double)p.scale;
}

// ---- vector paths ---------------------------------------------------------------------------------------------
// CL = true : channels-last / [B,C] (step_b == 1, size_b % VEC == 0): one bias vector, one noise scalar per lane
// CL = false: planar NCHW (step_b % VEC == 0): one bias scalar, one noise vector per lane
template <typename T, bool CL>
__global__ __launch_bounds__(256) void bias_act_vec_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                           const T* __restrict__ ref, T* __restrict__ y,
                                                           const float* __restrict__ noise,
                                                           const float* __restrict__ noise_w, BiasActParams p) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const long long nvec = p.size_x / VEC;
    const float nw = noise ? noise_w[0] : 0.f;
    for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
        const long long i = vi * VEC;
        V v, r, o;
        v.raw = *reinterpret_cast<const uint4*>(x + i);
        const bool use_ref = (p.grad == 1) && ref;
        if (use_ref) r.raw = *reinterpret_cast<const uint4*>(ref + i);
        float f[VEC], add[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) add[e] = 0.f;
        if (CL) {
            const long long q = i / p.size_b;                    // pixel (or row) index
            const int c0 = (int)(i - q * p.size_b);
            if (noise) {
                const float nv = nw * noise[p.noise_batch == 1 ? q % p.pix : q];
#pragma unroll
                for (int e = 0; e < VEC; ++e) add[e] = nv;
            }
            if (bias) {
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + c0 + e);
                    add[e] += b4.x; add[e + 1] += b4.y; add[e + 2] += b4.z; add[e + 3] += b4.w;
                }
            }
        } else {
            const long long q = i / p.step_b;                    // plane index b*C + c
            const int off = (int)(i - q * p.step_b);
            if (noise) {
                const long long base = (p.noise_batch == 1 ? 0 : (q / p.size_b) * p.pix) + off;
#pragma unroll
                for (int e = 0; e < VEC; ++e) add[e] = nw * noise[base + e];
            }
            if (bias) {
                const float bv = bias[q % p.size_b];
#pragma unroll
                for (int e = 0; e < VEC; ++e) add[e] += bv;
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float val = v.get(e) + add[e];
            f[e] = (p.grad == 2) ? 0.f : act_apply(val, use_ref ? r.get(e) : val, p.act, p.alpha, p.scale);
        }
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, f[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(y + i) = o.raw;
    }
}

template <typename T>
static int fwd_dispatch(const void* x, const float* bias, const void* ref, void* y, const float* noise,
                        const float* noise_w, const BiasActParams& p, hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)ref) & 15u) == 0 &&
                         (((uintptr_t)bias) & 15u) == 0;
    const bool cl = p.step_b == 1 && p.size_b % VEC == 0;
    const bool planar = p.step_b % VEC == 0 && (!noise || p.pix == p.step_b);
    if (aligned && p.size_x % VEC == 0 && (cl || planar)) {
        const long long nvec = p.size_x / VEC;
        const unsigned blocks = (unsigned)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
        if (cl)
            hipLaunchKernelGGL((bias_act_vec_kernel<T, true>), dim3(blocks), dim3(256), 0, s, (const T*)x, bias,
                               (const T*)ref, (T*)y, noise, noise_w, p);
        else
            hipLaunchKernelGGL((bias_act_vec_kernel<T, false>), dim3(blocks), dim3(256), 0, s, (const T*)x, bias,
                               (const T*)ref, (T*)y, noise, noise_w, p);
    } else {
        const long long blocks = (p.size_x + 255) / 256;
        if (blocks >= (1ll << 31)) return MSG_EUNSUPPORTED;
        hipLaunchKernelGGL((bias_act_scalar_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)x, bias,
                           (const T*)ref, (T*)y, noise, noise_w, p);
    }
    return MSG_CHECK_LAUNCH();
}

extern "C" int msg_fused_bias_act(const void* x, const float* bias, const void* ref, void* y, int dtype,
                                  long long size_x, int step_b, int size_b,
                                  const float* noise, const float* noise_weight, int noise_batch, int pix,
                                  int act, int grad, float alpha, float scale, void* stream) {
    if (size_x == 0) return MSG_OK;
    if (!x || !y || size_x < 0 || step_b <= 0 || size_b <= 0 || (act != 1 && act != 3) || grad < 0 || grad > 2)
        return MSG_EINVAL;
    if (noise && (!noise_weight || pix <= 0 || noise_batch <= 0)) return MSG_EINVAL;
    if (grad == 1 && act == 3 && !ref) return MSG_EINVAL;
    if (size_x == 0) return MSG_OK;
    BiasActParams p{size_x, step_b, size_b, noise_batch, pix, act, grad, alpha, scale};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_F32) return fwd_dispatch<float>(x, bias, ref, y, noise, noise_weight, p, s);
    if (dtype == MSG_BF16) return fwd_dispatch<bf16_t>(x, bias, ref, y, noise, noise_weight, p, s);
    if (dtype == MSG_F16) return fwd_dispatch<f16_t>(x, bias, ref, y, noise, noise_weight, p, s);
    if (dtype == MSG_F64) {                          // `bias` holds float64 values (the reference's bias has the input's dtype)
        if (noise) return MSG_EUNSUPPORTED;
        const long long blocks = (size_x + 255) / 256;
        if (blocks >= (1ll << 31)) return MSG_EUNSUPPORTED;
        hipLaunchKernelGGL(bias_act_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const double*)x,
                           (const double*)(const void*)bias, (const double*)ref, (double*)y, p);
        return MSG_CHECK_LAUNCH();
    }
    return MSG_EUNSUPPORTED;
}

// ---- backward with reductions ---------------------------------------------------------------------------------
// grad_bias[c] and grad_noise_weight are sums over the whole map.  They are NOT accumulated with float atomics (arrival
// order = run-to-run differences in every training step): each workgroup stores its partial sums in a workspace,
// part_b[j][c] (j = the workgroup's slice of the pixels) and part_n[block], and bias_act_bwd_reduce_kernel adds them in
// index order.  Same shapes -> same grid -> same association -> bit-identical results.
//
// Channels-last: block = LC channel-vectors x (256/LC) pixel lanes; every lane keeps its VEC channels across its
// pixel loop, partial sums meet in LDS and are added there in lane order.
// MASK: `out` is not read; the sign of the 8 outputs of a channel vector comes from one byte of the map the forward kernel
// wrote beside its output (ActEpilogue::mask, msg_common.h): 2 + 1/16 instead of 3 passes' worth of traffic.
// SUMS_ONLY: nothing but the channel sums of gy (the bias gradient of a conv without an activation behind it).
template <typename T, bool HAS_NOISE, bool MASK = false, bool SUMS_ONLY = false>
__global__ __launch_bounds__(256) void bias_act_bwd_cl_kernel(const T* __restrict__ gy, const T* __restrict__ out,
                                                              T* __restrict__ gx, float* __restrict__ part_b,
                                                              const float* __restrict__ noise,
                                                              float* __restrict__ part_n, BiasActParams p,
                                                              int lanes_c, long long npix, long long pix_per_block,
                                                              int tile_m = 1, int tile_n = 8) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    __shared__ float red[256 * VEC + 256];
    const int lc = threadIdx.x % lanes_c, pl = threadIdx.x / lanes_c, npl = 256 / lanes_c;
    const int cv = blockIdx.x * lanes_c + lc;
    const long long p0 = (long long)blockIdx.y * pix_per_block;
    const long long p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
    float sb[VEC], sn = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) sb[e] = 0.f;
    // Two pixels per trip, both requested before either is used: with one 16-byte load in flight per thread the kernel sat at
    // 4.7 TB/s (32 KiB in flight per CU; Little's law wants ~48 KiB at this latency).  The sums still add pixel q before pixel
    // q + npl: the association -- and with it every bit of the bias / noise gradients -- is the one-pixel loop's.
    for (long long q = p0 + pl; q < p1; q += 2 * npl) {
        // (no fused multiply-adds here: f[e] is stored AND summed; a product folded into the running sum in one instantiation
        //  and not in the other would make the masked and the unmasked kernel differ in the sums' last bit)
#pragma clang fp contract(off)
        const long long qq[2] = {q, q + npl};
        const bool live1 = qq[1] < p1;
        V g[2], o[2];
        unsigned int mbits[2] = {0u, 0u};
        float nz[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !live1) break;
            const long long i = qq[u] * p.size_b + (long long)cv * VEC;
            g[u].raw = *reinterpret_cast<const uint4*>(gy + i);
            if constexpr (!SUMS_ONLY) {
                if constexpr (MASK) mbits[u] = reinterpret_cast<const unsigned char*>(out)[act_mask_index(qq[u], cv, p.size_b, tile_m, tile_n)];
                else o[u].raw = *reinterpret_cast<const uint4*>(out + i);
                if (HAS_NOISE) nz[u] = noise[p.noise_batch == 1 ? qq[u] % p.pix : qq[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !live1) break;
            if constexpr (SUMS_ONLY) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sb[e] += g[u].get(e);
                continue;
            }
            V r;
            float f[VEC], rowsum = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const bool pos = MASK ? ((mbits[u] >> e) & 1u) != 0 : o[u].get(e) > 0.f;
                f[e] = g[u].get(e) * p.scale * ((pos || p.act != 3) ? 1.f : p.alpha);
                sb[e] += f[e];
                rowsum += f[e];
            }
            if (HAS_NOISE) sn = fmaf(rowsum, nz[u], sn);
            if constexpr (VEC == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r.set(e, f[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r.set2(e, f[2 * e], f[2 * e + 1]);
            }
            *reinterpret_cast<uint4*>(gx + qq[u] * p.size_b + (long long)cv * VEC) = r.raw;
        }
    }
    if (part_b) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[(pl * lanes_c + lc) * VEC + e] = sb[e];
    }
    if (HAS_NOISE) red[256 * VEC + threadIdx.x] = sn;
    __syncthreads();
    if (part_b) {
        for (int j = threadIdx.x; j < lanes_c * VEC; j += 256) {
            float s = 0.f;
            for (int k = 0; k < npl; ++k) s += red[k * lanes_c * VEC + j];
            part_b[(long long)blockIdx.y * p.size_b + blockIdx.x * lanes_c * VEC + j] = s;
        }
    }
    if (HAS_NOISE && threadIdx.x < 64) {
        float s = red[256 * VEC + threadIdx.x] + red[256 * VEC + threadIdx.x + 64] +
                  red[256 * VEC + threadIdx.x + 128] + red[256 * VEC + threadIdx.x + 192];
        s = wave_sum(s);
        if (threadIdx.x == 0) part_n[(long long)blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
}

// Planar: grid.x = plane (b*C + c), grid.y = chunk of the plane; scalar loads, block reduction.
// part_b[(b * chunks + chunk)][c], part_n[plane * chunks + chunk].
template <typename T, bool HAS_NOISE>
__global__ __launch_bounds__(256) void bias_act_bwd_planar_kernel(const T* __restrict__ gy, const T* __restrict__ out,
                                                                  T* __restrict__ gx, float* __restrict__ part_b,
                                                                  const float* __restrict__ noise,
                                                                  float* __restrict__ part_n, BiasActParams p) {
    __shared__ float red[8];
    const long long plane = blockIdx.x;
    const int c = (int)(plane % p.size_b);
    const long long b = plane / p.size_b;
    const long long base = plane * p.step_b;
    const long long nbase = p.noise_batch == 1 ? 0 : b * p.pix;      // noise index of element j of this plane: b*pix + j
    float sb = 0.f, sn = 0.f;
    for (int j = blockIdx.y * 256 + threadIdx.x; j < p.step_b; j += gridDim.y * 256) {
        const float f = load_as_f32(gy + base + j) * p.scale *
                        ((load_as_f32(out + base + j) > 0.f || p.act != 3) ? 1.f : p.alpha);
        store_from_f32(gx + base + j, f);
        sb += f;
        if (HAS_NOISE) sn = fmaf(f, noise[nbase + j], sn);
    }
    sb = wave_sum(sb);
    if (HAS_NOISE) sn = wave_sum(sn);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = sb; red[4 + w] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (part_b) part_b[(b * gridDim.y + blockIdx.y) * p.size_b + c] = red[0] + red[1] + red[2] + red[3];
        if (HAS_NOISE) part_n[plane * gridDim.y + blockIdx.y] = red[4] + red[5] + red[6] + red[7];
    }
}

// Channels-last / [B, C] maps the vector kernel cannot take (channel count not a whole number of 16-byte vectors, or
// unaligned pointers): grid.x = channel, grid.y = slice of the pixels; a thread walks pixels q = y*256 + t, + 256*gridDim.y ...
template <typename T, bool HAS_NOISE>
__global__ __launch_bounds__(256) void bias_act_bwd_strided_kernel(const T* __restrict__ gy, const T* __restrict__ out,
                                                                   T* __restrict__ gx, float* __restrict__ part_b,
                                                                   const float* __restrict__ noise,
                                                                   float* __restrict__ part_n, BiasActParams p,
                                                                   long long npix) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    float sb = 0.f, sn = 0.f;
    for (long long q = (long long)blockIdx.y * 256 + threadIdx.x; q < npix; q += (long long)gridDim.y * 256) {
        const long long i = q * p.size_b + c;
        const float f = load_as_f32(gy + i) * p.scale * ((load_as_f32(out + i) > 0.f || p.act != 3) ? 1.f : p.alpha);
        store_from_f32(gx + i, f);
        sb += f;
        if (HAS_NOISE) sn = fmaf(f, noise[p.noise_batch == 1 ? q % p.pix : q], sn);
    }
    sb = wave_sum(sb);
    if (HAS_NOISE) sn = wave_sum(sn);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = sb; red[4 + w] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (part_b) part_b[(long long)blockIdx.y * p.size_b + c] = red[0] + red[1] + red[2] + red[3];
        if (HAS_NOISE) part_n[(long long)blockIdx.y * gridDim.x + c] = red[4] + red[5] + red[6] + red[7];
    }
}

// grad_bias[c] = sum_j part_b[j][c]: a block takes 8 channels x 32 slices of the partial rows (up to 1024 rows of partials
// behind the large maps: one thread per channel would walk them in a ~100 us dependent chain); every thread adds its
// contiguous slice in row order, the 32 slice sums meet in LDS and are added in slice order.  grad_nw: the LAST block adds
// the n_n partials, each thread its strided share in index order, then a fixed tree.  No atomics, fixed association.
constexpr int BRC = 8, BRJ = 32;
__global__ __launch_bounds__(256) void bias_act_bwd_reduce_kernel(const float* __restrict__ part_b, float* __restrict__ grad_bias,
                                                                  int C, long long n_b, const float* __restrict__ part_n,
                                                                  float* __restrict__ grad_nw, long long n_n, int bias_blocks) {
    __shared__ float red[256];
    if ((int)blockIdx.x < bias_blocks) {
        const int cl = threadIdx.x % BRC, jg = threadIdx.x / BRC;
        const int c = blockIdx.x * BRC + cl;
        const long long per = (n_b + BRJ - 1) / BRJ;
        const long long j0 = jg * per, j1 = (j0 + per < n_b) ? j0 + per : n_b;
        float s = 0.f;
        if (c < C) {
#pragma unroll 8
            for (long long j = j0; j < j1; ++j) s += part_b[j * C + c];
        }
        red[jg * BRC + cl] = s;
        __syncthreads();
        if (threadIdx.x < BRC && c < C) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < BRJ; ++g) t += red[g * BRC + threadIdx.x];
            grad_bias[c] = t;
        }
        return;
    }
    float s = 0.f;
    for (long long j = threadIdx.x; j < n_n; j += 256) s += part_n[j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *grad_nw = red[0];
}

// The second stage on its own: for partial sums another kernel left (the activation backward in a conv epilogue,
// conv_fprop_row3.hip): grad_bias[c] = sum over n_b rows of part_b[.][C], *grad_nw = sum of n_n entries of part_n.
extern "C" int msg_bias_act_reduce_launch(const float* part_b, float* grad_bias, int C, long long n_b, const float* part_n,
                                          float* grad_nw, long long n_n, void* stream) {
    const int bias_blocks = (part_b && grad_bias) ? (C + BRC - 1) / BRC : 0;
    const bool has_noise = part_n && grad_nw;
    if (bias_blocks + (has_noise ? 1 : 0) == 0) return MSG_OK;
    hipLaunchKernelGGL(bias_act_bwd_reduce_kernel, dim3(bias_blocks + (has_noise ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, part_b,
                       grad_bias, C, n_b, part_n, grad_nw, n_n, bias_blocks);
    return MSG_CHECK_LAUNCH();
}

// out[c] = sum over `rows` rows of part[.][cols] in index order (the fixed-order second stage above as an entry of its own: the
// style-gradient partials of msg_modulate_backward, which the caller summed with a stock reduction -- one more library launch
// and ~7 us of host time per styled layer and backward).
extern "C" int msg_sum_rows(const float* part, float* out, long long rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || !part || !out) return MSG_EINVAL;
    return msg_bias_act_reduce_launch(part, out, cols, rows, nullptr, nullptr, 0, stream);
}

// The launch geometry of the backward, a function of the SHAPE only (the workspace query and the launch must agree).
struct BwdPlan {
    int path;                   // 0 channels-last vectors, 1 planar, 2 strided
    int lanes_c, gx_blocks;     // path 0
    long long npix, ppb, gy_blocks;
    long long planes; int chunks;
    long long n_b, n_n;         // partial rows of part_b ([n_b][C]) / entries of part_n
};

static BwdPlan bwd_plan(long long size_x, int step_b, int size_b, int vec) {
    BwdPlan q{};
    if (step_b == 1 && size_b % vec == 0) {
        const int nvec = size_b / vec;
        int lanes_c = 1;
        while (lanes_c < 64 && nvec % (lanes_c * 2) == 0) lanes_c *= 2;
        q.path = 0;
        q.lanes_c = lanes_c;
        q.npix = size_x / size_b;
        q.gx_blocks = nvec / lanes_c;
        long long want_y = 1024 / q.gx_blocks; if (want_y < 1) want_y = 1;
        const long long npl = 256 / lanes_c;
        long long ppb = (q.npix + want_y - 1) / want_y;
        q.ppb = ((ppb + npl - 1) / npl) * npl;
        q.gy_blocks = (q.npix + q.ppb - 1) / q.ppb;
        q.n_b = q.gy_blocks;
        q.n_n = q.gy_blocks * q.gx_blocks;
    } else if (step_b == 1) {
        q.path = 2;
        q.npix = size_x / size_b;
        long long chunks = (q.npix + 4095) / 4096; if (chunks > 64) chunks = 64; if (chunks < 1) chunks = 1;
        q.chunks = (int)chunks;
        q.n_b = chunks;
        q.n_n = chunks * size_b;
    } else {
        q.path = 1;
        q.planes = size_x / step_b;
        int chunks = (step_b + 4095) / 4096; if (chunks > 64) chunks = 64; if (chunks < 1) chunks = 1;
        q.chunks = chunks;
        q.n_b = (q.planes / size_b) * chunks;
        q.n_n = q.planes * chunks;
    }
    return q;
}

template <typename T>
static int bwd_dispatch(const void* gy, const void* out, void* gx, float* grad_bias, const float* noise,
                        float* grad_nw, const BiasActParams& p, float* ws, long long ws_floats, hipStream_t s,
                        bool mask = false, int tile_m = 1, int tile_n = 8) {
    constexpr int VEC = Vec16<T>::N;
    const bool aligned = (((uintptr_t)gy | (mask ? 0 : (uintptr_t)out) | (uintptr_t)gx) & 15u) == 0;
    const bool has_noise = noise && grad_nw;
    BwdPlan q = bwd_plan(p.size_x, p.step_b, p.size_b, aligned ? VEC : (1 << 30));
    // workspace: [n_b][C] bias partials, then n_n noise partials
    const long long need_b = grad_bias ? q.n_b * p.size_b : 0, need_n = has_noise ? q.n_n : 0;
    if (need_b + need_n > 0 && (!ws || ws_floats < need_b + need_n)) return MSG_EINVAL;
    float* part_b = grad_bias ? ws : nullptr;
    float* part_n = has_noise ? ws + need_b : nullptr;
    if (mask && (q.path != 0 || VEC != 8)) return MSG_EUNSUPPORTED;     // sign bytes: one per 8-channel vector, channels-last
    if (q.path == 0 && mask) {
        dim3 grid(q.gx_blocks, (unsigned)q.gy_blocks);
        if (has_noise)
            hipLaunchKernelGGL((bias_act_bwd_cl_kernel<T, true, true>), grid, dim3(256), 0, s, (const T*)gy, (const T*)out,
                               (T*)gx, part_b, noise, part_n, p, q.lanes_c, q.npix, q.ppb, tile_m, tile_n);
        else
            hipLaunchKernelGGL((bias_act_bwd_cl_kernel<T, false, true>), grid, dim3(256), 0, s, (const T*)gy, (const T*)out,
                               (T*)gx, part_b, noise, part_n, p, q.lanes_c, q.npix, q.ppb, tile_m, tile_n);
    } else if (q.path == 0) {
        dim3 grid(q.gx_blocks, (unsigned)q.gy_blocks);
        if (has_noise)
            hipLaunchKernelGGL((bias_act_bwd_cl_kernel<T, true>), grid, dim3(256), 0, s, (const T*)gy, (const T*)out,
                               (T*)gx, part_b, noise, part_n, p, q.lanes_c, q.npix, q.ppb);
        else
            hipLaunchKernelGGL((bias_act_bwd_cl_kernel<T, false>), grid, dim3(256), 0, s, (const T*)gy, (const T*)out,
                               (T*)gx, part_b, noise, part_n, p, q.lanes_c, q.npix, q.ppb);
    } else if (q.path == 2) {
        if (p.size_b > 65535 * 32) return MSG_EUNSUPPORTED;
        dim3 grid(p.size_b, q.chunks);
        if (has_noise)
            hipLaunchKernelGGL((bias_act_bwd_strided_kernel<T, true>), grid, dim3(256), 0, s, (const T*)gy, (const T*)out,
                               (T*)gx, part_b, noise, part_n, p, q.npix);
        else
            hipLaunchKernelGGL((bias_act_bwd_strided_kernel<T, false>), grid, dim3(256), 0, s, (const T*)gy, (const T*)out,
                               (T*)gx, part_b, noise, part_n, p, q.npix);
    } else {
        // planar NCHW
        if (q.planes >= (1ll << 31)) return MSG_EUNSUPPORTED;
        dim3 grid((unsigned)q.planes, q.chunks);
        if (has_noise)
            hipLaunchKernelGGL((bias_act_bwd_planar_kernel<T, true>), grid, dim3(256), 0, s, (const T*)gy,
                               (const T*)out, (T*)gx, part_b, noise, part_n, p);
        else
            hipLaunchKernelGGL((bias_act_bwd_planar_kernel<T, false>), grid, dim3(256), 0, s, (const T*)gy,
                               (const T*)out, (T*)gx, part_b, noise, part_n, p);
    }
    if (MSG_CHECK_LAUNCH() != MSG_OK) return MSG_ELAUNCH;
    if (need_b + need_n == 0) return MSG_OK;
    const int bias_blocks = grad_bias ? (p.size_b + BRC - 1) / BRC : 0;
    hipLaunchKernelGGL(bias_act_bwd_reduce_kernel, dim3(bias_blocks + (has_noise ? 1 : 0)), dim3(256), 0, s, part_b, grad_bias,
                       p.size_b, q.n_b, part_n, grad_nw, q.n_n, bias_blocks);
    return MSG_CHECK_LAUNCH();
}

extern "C" long long msg_bias_act_backward_workspace(long long size_x, int step_b, int size_b, int has_noise) {
    if (size_x <= 0 || step_b <= 0 || size_b <= 0 || size_x % ((long long)step_b * size_b) != 0) return 0;
    // (whether the vector path is taken depends on the storage type's vector width and on pointer alignment, which this
    //  query does not see: the largest need over the candidates)
    long long need = 0;
    const int vecs[3] = {4, 8, 1 << 30};
    for (int k = 0; k < 3; ++k) {
        const BwdPlan q = bwd_plan(size_x, step_b, size_b, vecs[k]);
        const long long n = q.n_b * size_b + (has_noise ? q.n_n : 0);
        if (n > need) need = n;
    }
    return need;
}

extern "C" int msg_bias_act_backward(const void* gy, const void* out, void* gx, int dtype,
                                     long long size_x, int step_b, int size_b,
                                     float* grad_bias, const float* noise, float* grad_noise_weight,
                                     int noise_batch, int pix, float alpha, float scale,
                                     float* ws, long long ws_floats, void* stream) {
    if (size_x < 0 || step_b <= 0 || size_b <= 0) return MSG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (size_x == 0) {                                 // empty map: the sums are zeros (the results are overwritten, not accumulated)
        if (grad_bias && hipMemsetAsync(grad_bias, 0, sizeof(float) * size_b, s) != hipSuccess) return MSG_ELAUNCH;
        if (noise && grad_noise_weight && hipMemsetAsync(grad_noise_weight, 0, sizeof(float), s) != hipSuccess) return MSG_ELAUNCH;
        return MSG_OK;
    }
    if (!gy || !out || !gx) return MSG_EINVAL;
    if (size_x % ((long long)step_b * size_b) != 0) return MSG_EINVAL;
    if (noise && grad_noise_weight && (pix <= 0 || noise_batch <= 0)) return MSG_EINVAL;
    if (noise && grad_noise_weight && step_b > 1 && pix != step_b) return MSG_EINVAL;
    BiasActParams p{size_x, step_b, size_b, noise_batch, pix, 3, 1, alpha, scale};
    if (dtype == MSG_F32) return bwd_dispatch<float>(gy, out, gx, grad_bias, noise, grad_noise_weight, p, ws, ws_floats, s);
    if (dtype == MSG_BF16) return bwd_dispatch<bf16_t>(gy, out, gx, grad_bias, noise, grad_noise_weight, p, ws, ws_floats, s);
    if (dtype == MSG_F16) return bwd_dispatch<f16_t>(gy, out, gx, grad_bias, noise, grad_noise_weight, p, ws, ws_floats, s);
    return MSG_EUNSUPPORTED;
}

// msg_bias_act_backward for a channels-last bf16 map whose forward kernel left the sign bytes of its output
// (msg_conv2d_fprop_act_mask, msg_upfirdn2d_separable_act_mask): `mask`, size_x / 8 bytes in the producer's tile order
// (tile_m consecutive pixels x tile_n channels per tile; 1 x size_b = plain [pixel][size_b / 8]), replaces `out`.
extern "C" int msg_bias_act_backward_mask(const void* gy, const unsigned char* mask, int tile_m, int tile_n, void* gx, int dtype,
                                          long long size_x, int size_b,
                                          float* grad_bias, const float* noise, float* grad_noise_weight,
                                          int noise_batch, int pix, float alpha, float scale,
                                          float* ws, long long ws_floats, void* stream) {
    if (size_x <= 0 || size_b <= 0 || !gy || !mask || !gx || size_x % size_b) return MSG_EINVAL;
    if (dtype != MSG_BF16 || size_b % 8) return MSG_EUNSUPPORTED;
    if (tile_m <= 0 || tile_n <= 0 || tile_n % 8 || size_b % tile_n || (size_x / size_b) % tile_m) return MSG_EINVAL;
    if (noise && grad_noise_weight && (pix <= 0 || noise_batch <= 0)) return MSG_EINVAL;
    BiasActParams p{size_x, 1, size_b, noise_batch, pix, 3, 1, alpha, scale};
    return bwd_dispatch<bf16_t>(gy, mask, gx, grad_bias, noise, grad_noise_weight, p, ws, ws_floats, (hipStream_t)stream, true,
                                tile_m, tile_n);
}

// ---- the same backward with the data gradient of a FEW-CHANNEL 1x1 modulated head computed on the fly -------------------
// The output of a styled 3x3 layer of the generator feeds the next level AND the level's image head, a 1x1 modulated conv
// without demodulation to n_head <= 8 planes (multi_stylegan_generator.py:513-523).  Its gradient was: the head's data
// gradient written as a 512-channel map (a streaming kernel, write-bound), autograd's sum with the other consumer's gradient,
// then the activation backward over that sum -- at 256^2 x 512 channels three passes over a gigabyte for six planes of
// information.  Here the activation backward forms the head's contribution itself,
//     h[q][c] = wscale * style[b][c] * sum_o ghead[q][o] * whead[o][c]           (the per-sample 1x1 weights, never stored)
// and takes the other consumer's gradient `gy` (or none: the last level) as the second term:
//     gx = (gy + h) * scale * (out > 0 ? 1 : alpha), sums as above.
// h is a K = 8 contraction per (pixel, channel): on the vector ALU it made the pass compute-bound (64 FMAs per stored 16 bytes:
// 2.3 TB/s); the matrix cores are idle in this kernel, so ONE v_mfma_f32_32x32x16_bf16 (upper half of K zero) forms the
// 32 pixels x 32 channels of a step.  The accumulator's layout -- a lane holds 16 channels of ONE pixel in groups of four --
// is the wrong one for the map (a store instruction would touch 32 rows with 32 bytes each: measured, half the bandwidth), so
// the 32 x 64 fp32 patch of a wave's step goes through a wave-private LDS patch and the streaming part runs in the layout of
// the plain kernel: eight lanes x 16 bytes = one 128-byte line of a row, eight rows per instruction.
// A workgroup = 8 waves x 64 consecutive channels, walking its pixel slice 32 pixels at a time; the per-sample head weights
// (rounded to bf16 like the weights of the launch this replaces) are built once per workgroup in registers.
typedef __bf16 ba_bf16v8 __attribute__((ext_vector_type(8)));
constexpr int HEAD_PITCH = 68;                              // floats per patch row (64 + 4: rows 16 bytes apart in the banks)
template <bool HAS_NOISE, bool HAS_GY>
__global__ __launch_bounds__(512) void bias_act_bwd_head_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ ghead,
                                                                const float* __restrict__ whead, const float* __restrict__ style,
                                                                float wscale, int n_head, const unsigned char* __restrict__ mask,
                                                                bf16_t* __restrict__ gx, float* __restrict__ part_b,
                                                                const float* __restrict__ noise, float* __restrict__ part_n,
                                                                BiasActParams p, int pix_per_block, int tile_m, int tm_shift,
                                                                int tile_n) {
    __shared__ __attribute__((aligned(16))) float patch[8][32 * HEAD_PITCH];
    __shared__ float red_n[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;              // MFMA view: row / column lp, K half lh
    const int lr = lane >> 3, lv = lane & 7;               // streaming view: row lr of a pass, 16-byte vector lv of the wave's 64 channels
    const int C = p.size_b;
    const int c_wave = blockIdx.x * 512 + wid * 64;        // this wave's first channel
    const bool wave_live = c_wave < C;                     // (C % 64 == 0: a wave's channels are all inside or all outside)
    const long long p0 = (long long)blockIdx.y * pix_per_block;       // (a multiple of 32 pixels, inside ONE sample: the launcher checks)
    const int b = (int)(p0 / p.pix);
    float* hp = patch[wid];
    ba_bf16v8 afrag[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = c_wave + 32 * k + lp;                // MFMA row lp of block k carries this channel
        const float sc = (lh == 0 && c < C) ? style[(long long)b * C + c] * wscale : 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float wv = (lh == 0 && c < C && o < n_head) ? whead[(long long)o * C + c] * sc : 0.f;
            afrag[k][o] = __builtin_bit_cast(__bf16, f2bf(wv));
        }
    }
    // planes >= n_head of the head's gradient are padding and may hold anything (NaN x 0 is NaN inside the MFMA): cleared
    unsigned keep[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
        keep[d] = (lh == 0 ? ((2 * d < n_head ? 0x0000ffffu : 0u) | (2 * d + 1 < n_head ? 0xffff0000u : 0u)) : 0u);
    // Everything inside the slice is addressed with 32-bit offsets from the slice's own base pointers.  Sign bytes:
    // index = ((tm * (C / tile_n) + tn) * tile_m + r) * vpt + c for pixel tile tm, row r, channel vector tn * vpt + c -- a
    // per-pixel part and a part that depends on the lane's channel vector only.
    const int vpt = tile_n >> 3;
    const int cv = (c_wave >> 3) + lv;
    const int m_col = (cv / vpt) * tile_m * vpt + cv % vpt;
    const int m_tile_stride = (C / tile_n) * tile_m * vpt;
    const int q_first = (int)p0;                            // (fewer than 2^31 elements: the launcher checks)
    const int c_lane = c_wave + 8 * lv;
    const bf16_t* gy_s = HAS_GY ? gy + p0 * C + c_lane : nullptr;
    bf16_t* gx_s = gx + p0 * C + c_lane;
    const bf16_t* gh_s = ghead + p0 * 8;
    const float* nz_s = HAS_NOISE ? noise + (p.noise_batch == 1 ? p0 - (long long)b * p.pix : p0) : nullptr;
    float sb[8], sn = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sb[e] = 0.f;
    const float pos_s = p.scale, neg_s = p.scale * p.alpha;
    if (wave_live) {
        for (int t0 = 0; t0 < pix_per_block; t0 += 32) {
#pragma clang fp contract(off)
            u32x4 hraw = *reinterpret_cast<const u32x4*>(gh_s + (t0 + lp) * 8);
            u32x4 g[4];
            unsigned mb[4];
            float nz[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int ql = t0 + 8 * ps + lr, qg = q_first + ql;
                if constexpr (HAS_GY) g[ps] = *reinterpret_cast<const u32x4*>(gy_s + ql * C);
                const int tm = tm_shift >= 0 ? qg >> tm_shift : qg / tile_m;
                mb[ps] = mask[tm * m_tile_stride + (qg - tm * tile_m) * vpt + m_col];
                nz[ps] = HAS_NOISE ? nz_s[ql] : 0.f;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) hraw[d] &= keep[d];
            const ba_bf16v8 bfrag = __builtin_bit_cast(ba_bf16v8, hraw);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[k], bfrag, acc, 0, 0, 0);
#endif
                // lane (lp, lh): pixel lp, channels 32 k + 8 i + 4 lh + (0..3), i = 0..3
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4*>(hp + lp * HEAD_PITCH + 32 * k + 8 * i + 4 * lh) =
                        f32x4{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = 8 * ps + lr;
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(hp + row * HEAD_PITCH + 8 * lv);
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(hp + row * HEAD_PITCH + 8 * lv + 4);
                float f[8], rowsum = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float gsum = e < 4 ? h0[e & 3] : h1[e & 3];
                    if constexpr (HAS_GY) {
                        const unsigned w = g[ps][e >> 1];
                        gsum += (e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
                    }
                    f[e] = gsum * (((mb[ps] >> e) & 1u) ? pos_s : neg_s);
                    sb[e] += f[e];
                    rowsum += f[e];
                }
                if (HAS_NOISE) sn = fmaf(rowsum, nz[ps], sn);
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (uint32_t)f2bf(f[2 * e]) | ((uint32_t)f2bf(f[2 * e + 1]) << 16);
                *reinterpret_cast<u32x4*>(gx_s + (t0 + row) * C) = pk;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                  // (the next step's patch writes stay behind these reads)
        }
    }
    // channel sums: over the eight row lanes that share a channel vector (lane = 8 lr + lv), fixed order
    if (part_b && wave_live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = sb[e];
#pragma unroll
            for (int off = 32; off >= 8; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lr == 0) part_b[(long long)blockIdx.y * C + c_lane + e] = s;
        }
    }
    if (HAS_NOISE) {
        sn = wave_sum(sn);
        if (lane == 0) red_n[wid] = sn;
        __syncthreads();
        if (threadIdx.x == 0)
            part_n[(long long)blockIdx.y * gridDim.x + blockIdx.x] =
                ((red_n[0] + red_n[1]) + (red_n[2] + red_n[3])) + ((red_n[4] + red_n[5]) + (red_n[6] + red_n[7]));
    }
}

// gy: the other consumer's gradient, bf16 [pixels][size_b], or NULL; ghead: the head's gradient, bf16 [pixels][8] (planes
// n_head .. 7 are padding and may hold anything); whead fp32
// [n_head][size_b]; style fp32 [B][size_b]; `pix` pixels per sample.  Workspace as msg_bias_act_backward_workspace(size_x, 1,
// size_b, noise != NULL).  size_b a multiple of 64, pix a multiple of 32 (MSG_EUNSUPPORTED otherwise).
extern "C" int msg_bias_act_backward_mask_head(const void* gy, const void* ghead, const float* whead, const float* style,
                                               float wscale, int n_head, const unsigned char* mask, int tile_m, int tile_n,
                                               void* gx, int dtype, long long size_x, int size_b,
                                               float* grad_bias, const float* noise, float* grad_noise_weight,
                                               int noise_batch, int pix, float alpha, float scale,
                                               float* ws, long long ws_floats, void* stream) {
    if (size_x <= 0 || size_b <= 0 || !ghead || !whead || !style || !mask || !gx || size_x % size_b || pix <= 0) return MSG_EINVAL;
    if (dtype != MSG_BF16 || size_b % 64 || pix % 32 || n_head < 1 || n_head > 8) return MSG_EUNSUPPORTED;
    if (tile_m <= 0 || tile_n <= 0 || tile_n % 8 || size_b % tile_n || (size_x / size_b) % tile_m) return MSG_EINVAL;
    if (noise && grad_noise_weight && noise_batch <= 0) return MSG_EINVAL;
    if ((((uintptr_t)gy | (uintptr_t)ghead | (uintptr_t)gx) & 15u) != 0) return MSG_EUNSUPPORTED;
    const long long npix = size_x / size_b;
    if (npix % pix) return MSG_EINVAL;
    if (size_x >= (1ll << 31)) return MSG_EUNSUPPORTED;                  // 32-bit offsets inside a pixel slice, 32-bit pixel index
    BiasActParams p{size_x, 1, size_b, noise_batch, pix, 3, 1, alpha, scale};
    const bool has_noise = noise && grad_noise_weight;
    // pixel slices: the plan of the plain backward (whose workspace the caller allocated), its slice rounded up to whole
    // 32-pixel steps that divide a sample -- never more slices than that plan has
    const BwdPlan q = bwd_plan(size_x, 1, size_b, 8);
    if (q.path != 0) return MSG_EUNSUPPORTED;
    long long ppb = ((q.ppb + 31) / 32) * 32;
    while (ppb < pix && pix % ppb) ppb += 32;
    if (ppb > pix) ppb = pix;
    const long long gy_blocks = npix / ppb;
    const int gx_blocks = (size_b + 511) / 512;
    if (gy_blocks > 65535 * 32ll || gy_blocks > q.gy_blocks) return MSG_EUNSUPPORTED;
    const long long need_b = grad_bias ? gy_blocks * size_b : 0, need_n = has_noise ? gy_blocks * gx_blocks : 0;
    if (need_b + need_n > 0 && (!ws || ws_floats < need_b + need_n)) return MSG_EINVAL;
    float* part_b = grad_bias ? ws : nullptr;
    float* part_n = has_noise ? ws + need_b : nullptr;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(gx_blocks, (unsigned)gy_blocks);
    int tm_shift = 0;
    while ((1 << tm_shift) < tile_m) ++tm_shift;
    if ((1 << tm_shift) != tile_m) tm_shift = -1;
#define HEAD_LAUNCH(N_, G_)                                                                                             \
    hipLaunchKernelGGL((bias_act_bwd_head_kernel<N_, G_>), grid, dim3(512), 0, s, (const bf16_t*)gy, (const bf16_t*)ghead, \
                       whead, style, wscale, n_head, mask, (bf16_t*)gx, part_b, noise, part_n, p, (int)ppb, tile_m, tm_shift, tile_n)
    if (has_noise) { if (gy) HEAD_LAUNCH(true, true); else HEAD_LAUNCH(true, false); }
    else { if (gy) HEAD_LAUNCH(false, true); else HEAD_LAUNCH(false, false); }
#undef HEAD_LAUNCH
    if (MSG_CHECK_LAUNCH() != MSG_OK) return MSG_ELAUNCH;
    if (need_b + need_n == 0) return MSG_OK;
    const int bias_blocks = grad_bias ? (size_b + BRC - 1) / BRC : 0;
    hipLaunchKernelGGL(bias_act_bwd_reduce_kernel, dim3(bias_blocks + (has_noise ? 1 : 0)), dim3(256), 0, s, part_b, grad_bias,
                       size_b, gy_blocks, part_n, grad_noise_weight, gy_blocks * gx_blocks, bias_blocks);
    return MSG_CHECK_LAUNCH();
}

// ---- y = (a + beta * b) * gain : the residual merges of the discriminator blocks ((main + residual) / sqrt(2),
// u_net_2d_discriminator.py:185,381) in one pass instead of an add and a mul --------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void scaled_add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                         long long nvec, float beta, float gain) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
        V va, vb, o;
        va.raw = *reinterpret_cast<const uint4*>(a + vi * VEC);
        vb.raw = *reinterpret_cast<const uint4*>(b + vi * VEC);
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = fmaf(beta, vb.get(e), va.get(e)) * gain;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, f[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(y + vi * VEC) = o.raw;
    }
}

extern "C" int msg_scaled_add(const void* a, const void* b, void* y, int dtype, long long n, float beta, float gain,
                              void* stream) {
    if (n == 0) return MSG_OK;
    if (!a || !b || !y || n < 0) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (n % vec || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15u)) return MSG_EUNSUPPORTED;
    const long long nvec = n / vec;
    const unsigned blocks = (unsigned)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((scaled_add_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, nvec, beta, gain);
    else
        hipLaunchKernelGGL((scaled_add_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, nvec, beta, gain);
    return MSG_CHECK_LAUNCH();
}

// Same, for operands that are channel-slices of channels-last buffers: [rows][cols] with row pitches lda / ldb / ldy
// (elements).  This is what the gradient of a skip connection looks like (a slice of the concatenated map's gradient).
template <typename T>
__global__ __launch_bounds__(256) void scaled_add_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                              long long rows, int cvecs, long long lda, long long ldb,
                                                              long long ldy, float beta, float gain) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const long long nvec = rows * cvecs;
    for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
        const long long r = vi / cvecs;
        const int c = (int)(vi - r * cvecs) * VEC;
        V va, vb, o;
        va.raw = *reinterpret_cast<const uint4*>(a + r * lda + c);
        vb.raw = *reinterpret_cast<const uint4*>(b + r * ldb + c);
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = fmaf(beta, vb.get(e), va.get(e)) * gain;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, f[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(y + r * ldy + c) = o.raw;
    }
}

extern "C" int msg_scaled_add_rows(const void* a, const void* b, void* y, int dtype, long long rows, int cols,
                                   long long lda, long long ldb, long long ldy, float beta, float gain, void* stream) {
    if (rows == 0 || cols == 0) return MSG_OK;
    if (!a || !b || !y || rows < 0 || cols < 0 || lda < cols || ldb < cols || ldy < cols) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (cols % vec || lda % vec || ldb % vec || ldy % vec || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15u))
        return MSG_EUNSUPPORTED;
    const long long nvec = rows * (cols / vec);
    const unsigned blocks = (unsigned)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((scaled_add_rows_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b,
                           (bf16_t*)y, rows, cols / vec, lda, ldb, ldy, beta, gain);
    else
        hipLaunchKernelGGL((scaled_add_rows_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)a, (const float*)b,
                           (float*)y, rows, cols / vec, lda, ldb, ldy, beta, gain);
    return MSG_CHECK_LAUNCH();
}

// ---- y = (gamma * a + b) * gain with gamma a DEVICE scalar: the merge of the NonLocalBlock, (gamma * o + residual) / sqrt(2)
// (u_net_2d_discriminator.py:381; gamma is a learnt fp32 parameter), and its backward in one pass:
//   ga = gamma * gain * gy,  gb = gain * gy,  g_gamma = gain * sum(gy * a)   (block partials + fixed-order second stage).
// The composite it replaces is five elementwise launches and a reduction over the same maps.
template <typename T>
__global__ __launch_bounds__(256) void gamma_merge_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              const float* __restrict__ gamma, T* __restrict__ y, long long nvec,
                                                              float gain) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    const float gm = gamma[0];
    for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
        V va, vb, o;
        va.raw = *reinterpret_cast<const uint4*>(a + vi * VEC);
        vb.raw = *reinterpret_cast<const uint4*>(b + vi * VEC);
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = fmaf(gm, va.get(e), vb.get(e)) * gain;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, f[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(y + vi * VEC) = o.raw;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gamma_merge_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ a,
                                                              const float* __restrict__ gamma, T* __restrict__ ga,
                                                              T* __restrict__ gb, float* __restrict__ part, long long nvec,
                                                              float gain) {
    using V = Vec16<T>;
    constexpr int VEC = V::N;
    __shared__ float red[4];
    const float gm = gamma[0] * gain;
    float acc = 0.f;
    for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
        V vg, va, oa, ob;
        vg.raw = *reinterpret_cast<const uint4*>(gy + vi * VEC);
        va.raw = *reinterpret_cast<const uint4*>(a + vi * VEC);
        float fa[VEC], fb[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float g = vg.get(e);
            fa[e] = g * gm;
            fb[e] = g * gain;
            acc = fmaf(g, va.get(e), acc);
        }
        if constexpr (VEC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { oa.set(e, fa[e]); ob.set(e, fb[e]); }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { oa.set2(e, fa[2 * e], fa[2 * e + 1]); ob.set2(e, fb[2 * e], fb[2 * e + 1]); }
        }
        *reinterpret_cast<uint4*>(ga + vi * VEC) = oa.raw;
        *reinterpret_cast<uint4*>(gb + vi * VEC) = ob.raw;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void gamma_merge_reduce_kernel(const float* __restrict__ part, int n, float gain,
                                                                 float* __restrict__ g_gamma) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];            // (fixed assignment, fixed order: deterministic)
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) g_gamma[0] = red[0] * gain;
}

constexpr int GAMMA_MERGE_BLOCKS = 2048;

extern "C" int msg_gamma_merge(const void* a, const void* b, const float* gamma, void* y, int dtype, long long n, float gain,
                               void* stream) {
    if (n == 0) return MSG_OK;
    if (!a || !b || !gamma || !y || n < 0) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (n % vec || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15u)) return MSG_EUNSUPPORTED;
    const long long nvec = n / vec;
    const unsigned blocks = (unsigned)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((gamma_merge_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, gamma, (bf16_t*)y, nvec, gain);
    else
        hipLaunchKernelGGL((gamma_merge_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)a, (const float*)b, gamma, (float*)y, nvec, gain);
    return MSG_CHECK_LAUNCH();
}

// ws: at least msg_gamma_merge_backward_workspace() floats (contents irrelevant); g_gamma [1] fp32, overwritten.
extern "C" long long msg_gamma_merge_backward_workspace(void) { return GAMMA_MERGE_BLOCKS; }

extern "C" int msg_gamma_merge_backward(const void* gy, const void* a, const float* gamma, void* ga, void* gb, float* g_gamma,
                                        int dtype, long long n, float gain, float* ws, void* stream) {
    if (!gy || !a || !gamma || !ga || !gb || !g_gamma || !ws || n <= 0) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (n % vec || (((uintptr_t)gy | (uintptr_t)a | (uintptr_t)ga | (uintptr_t)gb) & 15u)) return MSG_EUNSUPPORTED;
    const long long nvec = n / vec;
    const unsigned blocks = (unsigned)((nvec + 255) / 256 < GAMMA_MERGE_BLOCKS ? (nvec + 255) / 256 : GAMMA_MERGE_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((gamma_merge_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)gy, (const bf16_t*)a, gamma, (bf16_t*)ga, (bf16_t*)gb, ws, nvec, gain);
    else
        hipLaunchKernelGGL((gamma_merge_bwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)gy, (const float*)a, gamma, (float*)ga, (float*)gb, ws, nvec, gain);
    if (hipGetLastError() != hipSuccess) return MSG_ELAUNCH;
    hipLaunchKernelGGL(gamma_merge_reduce_kernel, dim3(1), dim3(256), 0, s, ws, (int)blocks, gain, g_gamma);
    return MSG_CHECK_LAUNCH();
}

// sums[c] = sum over the pixels of a channels-last map x [size_x / C][C] (fp32, overwritten; deterministic: workgroup partials in
// ws -- msg_bias_act_backward_workspace(size_x, 1, C, 0) floats -- and the fixed-order second stage of the activation backward).
// The bias gradient of a convolution that has no activation behind it (the discriminator's strided convs).
extern "C" int msg_channel_sums(const void* x, float* sums, int dtype, long long size_x, int C, float* ws, long long ws_floats,
                                void* stream) {
    if (size_x <= 0 || C <= 0 || !x || !sums || !ws || size_x % C) return MSG_EINVAL;
    if (dtype != MSG_F32 && dtype != MSG_BF16) return MSG_EUNSUPPORTED;
    const int vec = dtype == MSG_BF16 ? 8 : 4;
    if (C % vec || ((uintptr_t)x & 15u)) return MSG_EUNSUPPORTED;
    const BwdPlan q = bwd_plan(size_x, 1, C, vec);
    if (q.path != 0 || ws_floats < q.n_b * C) return MSG_EINVAL;
    BiasActParams p{size_x, 1, C, 1, 1, 3, 1, 0.f, 1.f};
    dim3 grid(q.gx_blocks, (unsigned)q.gy_blocks);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSG_BF16)
        hipLaunchKernelGGL((bias_act_bwd_cl_kernel<bf16_t, false, false, true>), grid, dim3(256), 0, s, (const bf16_t*)x,
                           (const bf16_t*)nullptr, (bf16_t*)nullptr, ws, (const float*)nullptr, (float*)nullptr, p, q.lanes_c,
                           q.npix, q.ppb, 1, 8);
    else
        hipLaunchKernelGGL((bias_act_bwd_cl_kernel<float, false, false, true>), grid, dim3(256), 0, s, (const float*)x,
                           (const float*)nullptr, (float*)nullptr, ws, (const float*)nullptr, (float*)nullptr, p, q.lanes_c,
                           q.npix, q.ppb, 1, 8);
    if (hipGetLastError() != hipSuccess) return MSG_ELAUNCH;
    const int bias_blocks = (C + BRC - 1) / BRC;
    hipLaunchKernelGGL(bias_act_bwd_reduce_kernel, dim3(bias_blocks), dim3(256), 0, s, ws, sums, C, q.n_b, (const float*)nullptr,
                       (float*)nullptr, 0ll, bias_blocks);
    return MSG_CHECK_LAUNCH();
}
