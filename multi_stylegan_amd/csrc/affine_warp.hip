// Batched affine warp for adaptive discriminator augmentation: the arithmetic the reference delegates to
// kornia 0.4.1 (kaf.rotate / kaf.apply_affine, multi_stylegan/adaptive_discriminator_augmentation.py:124,133,152,168,187)
// -- rotation-scale matrix about a centre, conjugation with the pixel -> [-1, 1] normalisation, inversion, affine_grid,
// bilinear grid_sample with zeros or reflection padding -- as ONE launch per stage over the whole batch, with the
// per-image selection (u[b] <= p, or <= 1 - sqrt(1 - p)) evaluated on the device from a device-resident p: no host
// synchronisation, no index lists, unselected images are copied through.  HBM-bound: one read of the 4 neighbours per
// channel (L2-served) and one coalesced write per output element.  Backward = the transpose of the bilinear gather, a
// scatter -- made DETERMINISTIC by accumulating in 64-bit fixed point (integer addition is associative: the order in which
// the atomics arrive cannot change the sum) and converting once at the end.  Parity unpinned (kornia is not available): the
// checker is oracle/ada.py, which runs the same published algorithm through torch's own affine_grid / grid_sample.
#include "msg_common.h"

namespace {

struct WarpParams {
    int B, C, H, W;
    float cx, cy;            // rotation / scaling centre in pixels (x, y)
    float angle_const;       // used when no per-image angle array is given (degrees)
    int padding;             // 0 zeros, 2 reflection (kornia SamplePadding)
    int align_corners;
    int rot_prob;            // selection threshold: 0 -> p, 1 -> 1 - sqrt(1 - p)
};

// torch's reflect_coordinates for align_corners = true (bounds [0, size - 1]) / false ([-0.5, size - 0.5]), then clip
__device__ __forceinline__ float reflect_coord(float v, int size, int align_corners) {
    const float lo2 = align_corners ? 0.f : -1.f, hi2 = align_corners ? 2.f * (size - 1) : 2.f * size - 1.f;
    if (lo2 == hi2) return 0.f;
    const float mn = lo2 * 0.5f, span = (hi2 - lo2) * 0.5f;
    v = fabsf(v - mn);
    const float extra = fmodf(v, span);
    const int flips = (int)floorf(v / span);
    v = (flips & 1) ? span - extra + mn : extra + mn;
    return fminf(fmaxf(v, 0.f), (float)(size - 1));
}

// source sampling position (pixels, torch grid_sample convention) of output pixel (xo, yo) of image b
__device__ __forceinline__ bool source_position(const WarpParams& p, const float* angle, const float* scale,
                                                const float* u, const float* prob, int b, int xo, int yo,
                                                float& xs, float& ys) {
    const float pr = prob[0];
    const float thr = p.rot_prob ? 1.f - sqrtf(1.f - pr) : pr;
    if (!(u[b] <= thr)) return false;
    const float deg = angle ? angle[b] : p.angle_const;
    const float sx = scale ? scale[2 * b] : 1.f, sy = scale ? scale[2 * b + 1] : 1.f;
    float sn, cs;
    sincosf(deg * 0.017453292519943295f, &sn, &cs);
    // kornia get_rotation_matrix2d: [[cos, sin], [-sin, cos]] @ diag(sx, sy), translation from the first row
    const float a = cs * sx, bq = sn * sy, c = -sn * sx, d = cs * sy;
    const float tx = (1.f - a) * p.cx - bq * p.cy, ty = bq * p.cx + (1.f - a) * p.cy;
    const float det = a * d - bq * c, inv = 1.f / det;
    const float ia = d * inv, ib = -bq * inv, ic = -c * inv, id = a * inv;          // inverse of the linear part
    const float itx = -(ia * tx + ib * ty), ity = -(ic * tx + id * ty);
    // affine_grid: normalised output coordinate; kornia's conjugation uses the [0, size - 1] <-> [-1, 1] map throughout
    const float nx = 2.f / (p.W - 1), ny = 2.f / (p.H - 1);
    const float un = p.align_corners ? 2.f * xo / (p.W - 1) - 1.f : (2.f * xo + 1.f) / p.W - 1.f;
    const float vn = p.align_corners ? 2.f * yo / (p.H - 1) - 1.f : (2.f * yo + 1.f) / p.H - 1.f;
    const float px = (un + 1.f) / nx, py = (vn + 1.f) / ny;                 // kornia pixel frame
    const float qx = ia * px + ib * py + itx, qy = ic * px + id * py + ity;
    const float sun = nx * qx - 1.f, svn = ny * qy - 1.f;                    // back to normalised
    xs = p.align_corners ? (sun + 1.f) * 0.5f * (p.W - 1) : ((sun + 1.f) * p.W - 1.f) * 0.5f;     // grid_sample unnormalise
    ys = p.align_corners ? (svn + 1.f) * 0.5f * (p.H - 1) : ((svn + 1.f) * p.H - 1.f) * 0.5f;
    if (p.padding == 2) {
        xs = reflect_coord(xs, p.W, p.align_corners);
        ys = reflect_coord(ys, p.H, p.align_corners);
    }
    return true;
}

// gradient contributions are accumulated as round(v * 2^38): resolution 3.6e-12 (image gradients of the mean-reduced losses
// are 1e-8 .. 1e-2), range +-3.3e7 per pixel.  RANGE ASSUMPTION, enforced: a single contribution must be finite and below
// FIX_LIMIT = 2^17 in magnitude (2^55 in fixed point: 256 such contributions to one pixel still cannot wrap the 64-bit sum);
// anything else raises the poison word behind the accumulators and the whole gradient comes out NaN.
constexpr double FIX_SCALE = 274877906944.0, FIX_INV = 1.0 / 274877906944.0;
constexpr float FIX_LIMIT = 131072.f;

template <bool BACKWARD>
__global__ __launch_bounds__(256) void affine_warp_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         unsigned long long* __restrict__ acc,
                                                         const float* __restrict__ angle, const float* __restrict__ scale,
                                                         const float* __restrict__ u, const float* __restrict__ prob,
                                                         WarpParams p) {
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long hw = (long long)p.H * p.W;
    if (pix >= (long long)p.B * hw) return;
    const int b = (int)(pix / hw);
    const int rem = (int)(pix - b * hw), yo = rem / p.W, xo = rem - yo * p.W;
    const float* ib = in + (long long)b * p.C * hw;
    float* ob = out + (long long)b * p.C * hw;
    float xs, ys;
    if (!source_position(p, angle, scale, u, prob, b, xo, yo, xs, ys)) {
        if (BACKWARD) return;                        // (copied through by affine_warp_finish_kernel)
        for (int c = 0; c < p.C; ++c) ob[c * hw + rem] = ib[c * hw + rem];
        return;
    }
    const float xf = floorf(xs), yf = floorf(ys);
    const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = xs - xf, wx0 = 1.f - wx1, wy1 = ys - yf, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < p.W, vx1 = x1 >= 0 && x1 < p.W, vy0 = y0 >= 0 && y0 < p.H, vy1 = y1 >= 0 && y1 < p.H;
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
    for (int c = 0; c < p.C; ++c) {
        if (!BACKWARD) {
            const float* s = ib + c * hw;
            float v = 0.f;
            if (vy0 && vx0) v += s[(long long)y0 * p.W + x0] * w00;
            if (vy0 && vx1) v += s[(long long)y0 * p.W + x1] * w01;
            if (vy1 && vx0) v += s[(long long)y1 * p.W + x0] * w10;
            if (vy1 && vx1) v += s[(long long)y1 * p.W + x1] * w11;
            ob[c * hw + rem] = v;
        } else {          // in = gy; acc = fixed-point gradient of x (zeroed): the transpose of the gather above
            const float g = ib[c * hw + rem];
            unsigned long long* s = acc + ((long long)b * p.C + c) * hw;
            unsigned long long* poison = acc + (long long)p.B * p.C * hw;
            auto add = [&](long long idx, float v) __attribute__((always_inline)) {
                // a non-finite or out-of-range contribution has no fixed-point image (the conversion is unspecified, the
                // sum would wrap into finite garbage): raise the poison word instead -- the finish kernel then writes NaN,
                // so a diverged gradient is still seen by the finite checks and the norm clipping downstream
                if (!(fabsf(v) < FIX_LIMIT)) { atomicOr(poison, 1ull); return; }
                atomicAdd(s + idx, (unsigned long long)__double2ll_rn((double)v * FIX_SCALE));   // two's complement wraps correctly
            };
            if (vy0 && vx0) add((long long)y0 * p.W + x0, g * w00);
            if (vy0 && vx1) add((long long)y0 * p.W + x1, g * w01);
            if (vy1 && vx0) add((long long)y1 * p.W + x0, g * w10);
            if (vy1 && vx1) add((long long)y1 * p.W + x1, g * w11);
        }
    }
}

// backward, second stage: gx = the accumulated gradient of a transformed image, gy itself for an image that was copied through
__global__ __launch_bounds__(256) void affine_warp_finish_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                const unsigned long long* __restrict__ acc,
                                                                const float* __restrict__ u, const float* __restrict__ prob,
                                                                WarpParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long chw = (long long)p.C * p.H * p.W;
    if (i >= (long long)p.B * chw) return;
    const int b = (int)(i / chw);
    const float pr = prob[0];
    const float thr = p.rot_prob ? 1.f - sqrtf(1.f - pr) : pr;
    const bool poisoned = acc[(long long)p.B * chw] != 0;
    gx[i] = (u[b] <= thr) ? (poisoned ? __builtin_nanf("") : (float)((double)(long long)acc[i] * FIX_INV)) : gy[i];
}

int launch(bool backward, const float* in, float* out, unsigned long long* acc, const float* angle, const float* scale,
           const float* u, const float* prob, const WarpParams& p, hipStream_t s) {
    const long long pixels = (long long)p.B * p.H * p.W;
    const unsigned blocks = (unsigned)((pixels + 255) / 256);
    if (!backward) {
        hipLaunchKernelGGL(affine_warp_kernel<false>, dim3(blocks), dim3(256), 0, s, in, out, acc, angle, scale, u, prob, p);
        return MSG_CHECK_LAUNCH();
    }
    const long long elems = pixels * p.C;
    if (hipMemsetAsync(acc, 0, sizeof(unsigned long long) * (elems + 1), s) != hipSuccess) return MSG_ELAUNCH;   // + poison word
    hipLaunchKernelGGL(affine_warp_kernel<true>, dim3(blocks), dim3(256), 0, s, in, out, acc, angle, scale, u, prob, p);
    if (MSG_CHECK_LAUNCH() != MSG_OK) return MSG_ELAUNCH;
    hipLaunchKernelGGL(affine_warp_finish_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, in, out, acc, u, prob, p);
    return MSG_CHECK_LAUNCH();
}

}  // namespace

extern "C" int msg_affine_warp(const float* x, float* y, const float* angle_deg, float angle_const, const float* scale_xy,
                               const float* select_u, const float* p, int rot_prob, float cx, float cy, int padding,
                               int align_corners, int B, int C, int H, int W, int backward, void* workspace,
                               void* stream) {
    if (B == 0) return MSG_OK;
    if (!x || !y || !select_u || !p || B < 0 || C <= 0 || H <= 1 || W <= 1) return MSG_EINVAL;
    if (backward && (!workspace || ((uintptr_t)workspace & 7u))) return MSG_EINVAL;
    if (padding != 0 && padding != 2) return MSG_EUNSUPPORTED;
    if ((long long)B * C * H * W >= (1ll << 40)) return MSG_EUNSUPPORTED;
    WarpParams wp{B, C, H, W, cx, cy, angle_const, padding, align_corners ? 1 : 0, rot_prob ? 1 : 0};
    return launch(backward != 0, x, y, (unsigned long long*)workspace, angle_deg, scale_xy, select_u, p, wp,
                  (hipStream_t)stream);
}
