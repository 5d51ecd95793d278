// crop_roi.hip -- image side of the CLIP distillation branch up to the tower's input (include/coda_clip_crops.h).
#include "coda_clip_crops.h"
#include "common.hip.h"

// every expression as written (float64 geometry that ends in int(): no FMA contraction)
#pragma clang fp contract(off)

namespace coda {
namespace {

struct ProjArgs {
  const float *corners, *sizes;
  const double *scale, *rot, *flip, *zx_flip, *kmat, *rtilt, *ori_wh, *offset_xy, *image_flip, *flip_length;
  double *uv, *depth;
  int32_t *rects;
  unsigned char *valid;
  int b, k;
};

__global__ __launch_bounds__(256) void project_rects_kernel(const ProjArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.b * a.k) return;
  const int bi = t / a.k;
  const float *c = a.corners + static_cast<size_t>(t) * 24;
  const double *sc = a.scale + bi * 3, *R = a.rot + bi * 9, *K = a.kmat + bi * 9, *T = a.rtilt + bi * 9;
  const double fx = a.flip[bi], fzx = a.zx_flip ? a.zx_flip[bi] : 1.0;
  const double wmax = a.ori_wh[bi * 2] - 1.0, hmax = a.ori_wh[bi * 2 + 1] - 1.0;
  const double offu = a.offset_xy[bi * 2], offv = a.offset_xy[bi * 2 + 1];
  const double ifl = a.image_flip[bi], fl = a.flip_length[bi];
  double umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY, dmin = INFINITY;
  for (int n = 0; n < 8; ++n) {
    // back out the point-cloud augmentation (:919-928): scale, rotation (row vector x matrix), flips
    const double g0 = static_cast<double>(c[n * 3]) * sc[0], g1 = static_cast<double>(c[n * 3 + 1]) * sc[1],
                 g2 = static_cast<double>(c[n * 3 + 2]) * sc[2];
    double p0 = g0 * R[0] + g1 * R[3] + g2 * R[6];
    double p1 = g0 * R[1] + g1 * R[4] + g2 * R[7];
    const double p2 = g0 * R[2] + g1 * R[5] + g2 * R[8];
    p1 = p1 * fzx;
    p0 = p0 * fx;
    // project (datasets/sunrgbd_utils.py:611-635): Rtilt^T p, depth -> camera axes (x, -z, y), K
    const double q0 = T[0] * p0 + T[3] * p1 + T[6] * p2;
    const double q1 = T[1] * p0 + T[4] * p1 + T[7] * p2;
    const double q2 = T[2] * p0 + T[5] * p1 + T[8] * p2;
    const double c0 = q0, c1 = -q2, c2 = q1;
    const double w0 = c0 * K[0] + c1 * K[1] + c2 * K[2];
    const double w1 = c0 * K[3] + c1 * K[4] + c2 * K[5];
    const double w2 = c0 * K[6] + c1 * K[7] + c2 * K[8];
    double u = w0 / (w2 + 1e-32), v = w1 / (w2 + 1e-32);
    // clip to the original image, move into the padded frame, undo the image flip (:949-965)
    u = fmin(fmax(u, 0.0), wmax) + offu;
    v = fmin(fmax(v, 0.0), hmax) + offv;
    u = u * ifl + (1.0 - ifl) * (fl - 1.0 - u);
    if (a.uv) {
      a.uv[(static_cast<size_t>(t) * 8 + n) * 2] = u;
      a.uv[(static_cast<size_t>(t) * 8 + n) * 2 + 1] = v;
    }
    if (a.depth) a.depth[static_cast<size_t>(t) * 8 + n] = w2;
    umin = fmin(umin, u); umax = fmax(umax, u);
    vmin = fmin(vmin, v); vmax = fmax(vmax, v);
    dmin = fmin(dmin, w2);
  }
  const int xmin = static_cast<int>(umin), ymin = static_cast<int>(vmin), xmax = static_cast<int>(umax),
            ymax = static_cast<int>(vmax);  // int(): truncation, :1021-1024
  a.rects[t * 4] = xmin; a.rects[t * 4 + 1] = ymin; a.rects[t * 4 + 2] = xmax; a.rects[t * 4 + 3] = ymax;
  const float smax = fmaxf(a.sizes[t * 3], fmaxf(a.sizes[t * 3 + 1], a.sizes[t * 3 + 2]));
  const bool ok = !(smax < 1e-16f) && (xmax - xmin) > 0 && (ymax - ymin) > 0 && !(dmin < 0.0);
  a.valid[t] = ok ? 1 : 0;
}

// PyTorch's cubic convolution coefficients (A = -0.75), aten/src/ATen/native/UpSample.h
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
  c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  c[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const unsigned char *__restrict__ images,
                                                          const int32_t *__restrict__ sel, const int32_t *__restrict__ rects,
                                                          const unsigned char *__restrict__ valid, float *__restrict__ out,
                                                          int h, int w, int k, int s, int res) {
  const int crop = blockIdx.y, bi = crop / s;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= res * res) return;
  const int oy = pix / res, ox = pix % res;
  const int box = sel[crop];
  const bool ok = box >= 0 && box < k && valid[bi * k + box] != 0;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  float *dst = out + static_cast<size_t>(crop) * 3 * res * res + pix;
  if (!ok) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) dst[static_cast<size_t>(ch) * res * res] = (1.0f - mean[ch]) / stdv[ch];
    return;
  }
  const int32_t *r = rects + (static_cast<size_t>(bi) * k + box) * 4;
  int xmin = r[0], ymin = r[1], xmax = r[2], ymax = r[3];
  // python slicing img[ymin:ymax, xmin:xmax] clips to the image (a valid rect lies inside the padded frame anyway)
  xmin = max(xmin, 0); ymin = max(ymin, 0); xmax = min(xmax, w); ymax = min(ymax, h);
  const int ch_ = ymax - ymin, cw_ = xmax - xmin;         // crop height / width ("w" / "h" in the reference, :1038-1039)
  const int edge = max(ch_, cw_);
  const int y_begin = (edge - ch_) / 2, x_begin = (edge - cw_) / 2;
  const float scale = static_cast<float>(edge) / static_cast<float>(res);
  const float sy = scale * (oy + 0.5f) - 0.5f, sx = scale * (ox + 0.5f) - 0.5f;
  const int iy = static_cast<int>(floorf(sy)), ix = static_cast<int>(floorf(sx));
  float cy[4], cx[4];
  cubic_coeffs(sy - iy, cy);
  cubic_coeffs(sx - ix, cx);
  const unsigned char *img = images + static_cast<size_t>(bi) * h * w * 3;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int yy = min(max(iy - 1 + i, 0), edge - 1) - y_begin;
    float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = min(max(ix - 1 + j, 0), edge - 1) - x_begin;
      const bool in = yy >= 0 && yy < ch_ && xx >= 0 && xx < cw_;
      const unsigned char *px = img + (static_cast<size_t>(ymin + yy) * w + (xmin + xx)) * 3;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) row[ch] += (in ? static_cast<float>(px[ch]) : 255.0f) * cx[j];
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) acc[ch] += row[ch] * cy[i];
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float q = rintf(fminf(fmaxf(acc[ch], 0.0f), 255.0f));  // clamp, round half to even, uint8
    dst[static_cast<size_t>(ch) * res * res] = (q / 255.0f - mean[ch]) / stdv[ch];
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_project_box_rects_f64(const float *corners, const float *sizes, const double *scale, const double *rot,
                                        const double *flip, const double *zx_flip, const double *kmat, const double *rtilt,
                                        const double *ori_wh, const double *offset_xy, const double *image_flip,
                                        const double *flip_length, double *uv, double *depth, int32_t *rects,
                                        unsigned char *valid, int b, int k, void *stream) {
  using namespace coda;
  if (b < 0 || k < 0) return CODA_EINVAL;
  if (b == 0 || k == 0) return CODA_OK;
  if (!corners || !sizes || !scale || !rot || !flip || !kmat || !rtilt || !ori_wh || !offset_xy || !image_flip ||
      !flip_length || !rects || !valid)
    return CODA_EINVAL;
  clear_sticky_error();
  const ProjArgs a{corners, sizes, scale, rot, flip, zx_flip, kmat, rtilt, ori_wh, offset_xy, image_flip, flip_length,
                   uv, depth, rects, valid, b, k};
  hipLaunchKernelGGL(project_rects_kernel, dim3((b * k + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return launch_status();
}

CODA_API int coda_crop_resize_f32(const unsigned char *images, const int32_t *sel, const int32_t *rects,
                                  const unsigned char *valid, float *out, int b, int h, int w, int k, int s, int res,
                                  void *stream) {
  using namespace coda;
  if (b < 0 || h <= 0 || w <= 0 || k <= 0 || s < 0 || res <= 0) return CODA_EINVAL;
  if (b == 0 || s == 0) return CODA_OK;
  if (!images || !sel || !rects || !valid || !out) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(crop_resize_kernel, dim3((res * res + 255) / 256, b * s), dim3(256), 0,
                     static_cast<hipStream_t>(stream), images, sel, rects, valid, out, h, w, k, s, res);
  return launch_status();
}
