// Shared device helpers for the gfx950 kernels (wave64, fp32 arithmetic, f32/bf16 storage).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/msg_hip.h"

// Kernel-selection and cost-model constants.  In the library that ships (and that bench.py measures) every one of them IS
// its default: `msg_tunable` is a constant expression there, the alternatives are dead code that the compiler drops, and
// the library reads no environment variable.  Only a -DMSG_TUNING build (the variant libraries of tools/, loaded through
// tools' MSG_LIB_VARIANT) reads MSG_* variables, for A/B measurements of alternatives whose verdicts DESIGN.md records.
#ifdef MSG_TUNING
static inline int msg_tunable(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
#define msg_tunable(name, dflt) (dflt)
#endif

typedef float  f32x4 __attribute__((ext_vector_type(4)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef short  bf16x8 __attribute__((ext_vector_type(8)));
typedef short  bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef _Float16 f16_t;            // IEEE half storage (the reference's `half` dispatch): FIR / activation entries only
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16-B staging register (plain vector: stays in VGPRs)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);      // RNE; keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
    return *reinterpret_cast<bf16_t*>(&h);
}

// 16-byte vector of storage elements <-> fp32 lanes.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    uint4 raw;
    __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ float get(int i) const { return __uint_as_float((&raw.x)[i]); }
    __device__ __forceinline__ void set(int i, float v) { (&raw.x)[i] = __float_as_uint(v); }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ float get(int i) const {
        uint32_t w = (&raw.x)[i >> 1];
        return (i & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
    }
    __device__ __forceinline__ void set2(int pair, float lo, float hi) {
        (&raw.x)[pair] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
};

template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ float get(int i) const {
        const uint32_t w = (&raw.x)[i >> 1];
        const unsigned short h = (unsigned short)((i & 1) ? (w >> 16) : (w & 0xffffu));
        return (float)__builtin_bit_cast(_Float16, h);
    }
    __device__ __forceinline__ void set2(int pair, float lo, float hi) {
        const unsigned short a = __builtin_bit_cast(unsigned short, (_Float16)lo);      // RNE
        const unsigned short b = __builtin_bit_cast(unsigned short, (_Float16)hi);
        (&raw.x)[pair] = (uint32_t)a | ((uint32_t)b << 16);
    }
};

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float load_as_f32<f16_t>(const f16_t* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void store_from_f32(T* p, float v);
template <> __device__ __forceinline__ void store_from_f32<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_from_f32<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// XCD-aware bijective block remap: blocks that share an XCD (bid % 8) get one
// contiguous chunk of the logical grid so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u;
    const unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Optional activation stage of the convolution epilogues: y = lrelu(conv + noise_w * noise[b, pixel] + bias[n]) * scale,
// applied to the ROUNDED conv result exactly as the stand-alone kernel (bias_act.hip) would apply it to the stored
// conv output -- the fused and the two-pass forms are bit-identical, only one read + one write of the map cheaper.
struct ActEpilogue {
    const float* bias;       // [N] or NULL
    const float* noise;      // [noise_batch][OH*OW] fp32 or NULL
    const float* noise_w;    // device scalar (with noise)
    int noise_batch;         // 1 (shared by the batch) or B
    int enabled;             // 0: off, 1: (noise +) bias + leaky ReLU, 2: residual merge y = (conv + residual) * res_gain
    float alpha, scale;
    const void* residual;    // enabled == 2: a map shaped like the output (same pixels), channel pitch res_ld elements
    int res_ld;
    float res_gain;
    unsigned char* mask;     // enabled == 1, bf16, optional: one byte per (pixel, 8-channel vector), bit e = (out[pixel][8 v + e] > 0):
                             // what the activation's backward needs of the output, at a sixteenth of its size (kernels that
                             // write it: conv_fprop_row3.hip, blur_sep.hip; the others ignore the field).  Layout: the
                             // producer's OUTPUT TILES one after the other (act_mask_index below), so that a workgroup writes
                             // one contiguous block instead of byte-granular pieces of lines it shares with other workgroups.
    // enabled == 3 (conv_fprop_row3.hip only; msg_conv2d_fprop_act_backward): this launch is the data gradient of the conv BEHIND an
    // activation and applies that activation's backward in its epilogue -- y = (conv [+ residual]) * (s > 0 ? scale : scale *
    // alpha), s = the activation's stored OUTPUT, known from `mask` (sign bytes another launch wrote in tiles of mask_tile_m x
    // mask_tile_n, act_mask_index) or from the output map itself (`sign_src`, channel pitch sign_ld) -- and leaves the partial sums
    // of the activation's bias / noise-weight gradients: part_b [rows][N] (one row per sample, pixel tile and wave row), part_n
    // (one value per sample, tile and wave; `noise` as for enabled == 1, without its weight).
    int mask_tile_m, mask_tile_n;
    const void* sign_src;
    int sign_ld;
    float* part_b;
    float* part_n;
};

// Byte index of (global pixel q, channel vector cv) in the sign-byte map of a [pixels][C] output produced in tiles of
// tile_m consecutive pixels x tile_n channels (tile_m = 1, tile_n = C: plain [pixel][C / 8]).
__device__ __host__ __forceinline__ long long act_mask_index(long long q, int cv, int C, int tile_m, int tile_n) {
    const int vpt = tile_n >> 3;                                 // vectors per tile row
    const long long tm = q / tile_m;
    const int r = (int)(q - tm * tile_m), tn = cv / vpt, c = cv - tn * vpt;
    return ((tm * (C / tile_n) + tn) * tile_m + r) * vpt + c;
}

// The sign byte of 8 stored bf16 outputs (see ActEpilogue::mask): "> 0" on the ROUNDED values, i.e. exactly what a
// backward pass reading the stored map tests -- a bf16 bit pattern is a positive number iff it is > 0 as a signed 16-bit
// integer (NaN payloads aside, as there).
__device__ __forceinline__ unsigned int act_sign_byte(u32x4 packed) {
    // Packed 16-bit arithmetic (13 instructions instead of ~30 compares / selects / shifts; in an epilogue that runs one wave
    // per SIMD each is four cycles): per half word, max(h, 0) as signed then min(., 1) as unsigned is 1 iff h > 0; the four
    // words' bits 0 / 16 are merged at bit 2k and the high halves folded down by 15.  Only the low byte is meaningful.
    // (inline assembly: written with the element-wise builtins the compiler recognises "h > 0" and goes back to compares)
    unsigned int t = 0;
    const unsigned int ones = 0x00010001u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned int h;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, %2" : "=&v"(h) : "v"((unsigned int)packed[k]), "s"(ones));
#else
        h = 0;
#endif
        t |= h << (2 * k);
    }
    return (t | (t >> 15)) & 0xffu;
}

// (conv + residual) * gain on VEC storage elements, the arithmetic of scaled_add_kernel (bias_act.hip): the residual
// merge of a discriminator block done in the epilogue of its 1x1 residual conv, bit-identical to the two-pass form.
template <typename T>
__device__ __forceinline__ u32x4 residual_epilogue_apply(u32x4 raw, u32x4 res, float gain) {
    constexpr int VEC = 16 / sizeof(T);
    Vec16<T> v, r, o;
    v.raw = make_uint4(raw[0], raw[1], raw[2], raw[3]);
    r.raw = make_uint4(res[0], res[1], res[2], res[3]);
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = fmaf(1.f, v.get(e), r.get(e)) * gain;
    if constexpr (VEC == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.set(e, f[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
    }
    u32x4 out;
    out[0] = o.raw.x; out[1] = o.raw.y; out[2] = o.raw.z; out[3] = o.raw.w;
    return out;
}

// `raw`: VEC storage elements of VEC consecutive channels of one pixel; `bv`: their bias values (0 where absent);
// `nv` = noise_w * noise[pixel] (or 0).  Same arithmetic, in the same order, as bias_act_vec_kernel.
// UNIT_SLOPE (the caller checked 0 <= alpha <= 1 on a uniform branch): the select is max(val, val * alpha) -- the same value
// for every input including zeros and NaN, two instructions fewer per element.
template <typename T, bool UNIT_SLOPE = false>
__device__ __forceinline__ u32x4 act_epilogue_apply(u32x4 raw, const float* bv, float nv, float alpha, float scale) {
    constexpr int VEC = 16 / sizeof(T);
    Vec16<T> v, o;
    v.raw = make_uint4(raw[0], raw[1], raw[2], raw[3]);
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float add = nv + bv[e];
        const float val = v.get(e) + add;
        f[e] = (UNIT_SLOPE ? fmaxf(val, val * alpha) : ((val > 0.f) ? val : val * alpha)) * scale;
    }
    if constexpr (VEC == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.set(e, f[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.set2(e, f[2 * e], f[2 * e + 1]);
    }
    u32x4 r;
    r[0] = o.raw.x; r[1] = o.raw.y; r[2] = o.raw.z; r[3] = o.raw.w;
    return r;
}

// ---- LDS-DMA as inline assembly (buffer_load_dwordx4 ... offen lds: lane L's 16 bytes land at LDS address m0 + 16 L).
// Through the builtins the compiler knows that the instruction writes LDS and orders every LDS read that MAY alias behind
// it with an s_waitcnt vmcnt(0) -- reads through ds_read_tr intrinsics always "may" -- which makes a multi-stage ring a queue
// of depth zero ("LDS-DMA for the weight gradient").  Here the compiler sees nothing and the waits are
// the kernel's own counted ones.  The descriptor is four SGPRs: base, no stride, 2^31 - 16 records, raw dword format;
// an offset of MSG_DMA_OOB (or any offset past the records) reads zeros.
typedef int msg_desc_t __attribute__((ext_vector_type(4)));
constexpr int MSG_DMA_OOB = (int)0x80000000;
__device__ __forceinline__ msg_desc_t msg_make_desc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    msg_desc_t d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    d[2] = 0x7ffffff0;
    d[3] = 0x00020000;
    return d;
}
// (m0 is named as clobbered so that the compiler never keeps a value of its own in it across a piece; clang warns that it
//  is a reserved register -- it has no other use for it in these kernels.  s_nop: one wait state between the scalar write of
//  m0 and the DMA that reads it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void msg_dma16(msg_desc_t desc, unsigned lds_addr, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory", "m0");
#endif
}
#pragma clang diagnostic pop

#define MSG_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? MSG_OK : MSG_ELAUNCH)
