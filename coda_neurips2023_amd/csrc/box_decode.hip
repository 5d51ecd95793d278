// box_decode.hip -- the box decoder of the 3DETR heads as one kernel per direction.
//
// Replaces the element-wise tail of get_box_predictions (models/model_3detr.py:1683-1731): sigmoid / offset /
// range normalisation of the centres (BoxProcessor.compute_predicted_center :64-70, utils/pc_util.py:38-66),
// sizes (:72-77), arg-max angle bin + residual + wrap (:79-95), the two corner builders
// (utils/box_util.py:383-424 depth frame, :427-490 camera frame through flip_axis_to_camera) and the
// objectness / class probabilities (:97-101).  In PyTorch this is ~70 launches of 3-8 us per direction on
// 16 384 rows (8 layers x 8 scenes x 256 queries) -- the host cannot enqueue them as fast as the GPU retires
// them, so the step stalled there for ~1.5 ms.  One thread owns a row; everything stays in registers.
#include "coda_box_ops.h"
#include "common.hip.h"

namespace coda {
namespace {

constexpr float kPi = 3.14159265358979323846f;
// corner sign patterns (utils/box_util.py:404-412 depth frame; :470-478 camera frame)
__constant__ float kSignXyz[8][3] = {{-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, -1}, {1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
__constant__ float kSignCam[8][3] = {{1, 1, 1}, {1, 1, -1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}, {-1, -1, 1}};

struct Strides {  // elements; per input: layer, scene, query (channels contiguous)
  long long v[5][3];
};
__device__ __forceinline__ const float *row_ptr(const float *base, const Strides &st, int which, int l, int bi, int q) {
  return base + l * st.v[which][0] + bi * st.v[which][1] + q * st.v[which][2];
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

struct Decoded {
  float sig_c[3], sig_s[3], cu[3], su[3], scale[3], inv_range[3];
  float angle, ca, sa;
  int bin;
};

__device__ __forceinline__ Decoded decode_row(const float *c_raw, const float *s_raw, const float *a_logit,
                                              const float *a_res, const float *qxyz, const float *lo, const float *hi,
                                              int nbin) {
  Decoded d;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    d.sig_c[a] = sigmoidf(c_raw[a]);
    d.sig_s[a] = sigmoidf(s_raw[a]);
    d.cu[a] = qxyz[a] + (d.sig_c[a] - 0.5f);
    const float range = hi[a] - lo[a];
    d.inv_range[a] = 1.0f / range;
    d.scale[a] = fmaxf(range, 0.1f);
    d.su[a] = d.sig_s[a] * d.scale[a];
  }
  if (nbin == 1) {  // datasets without rotation (:80-85)
    d.bin = 0;
    d.angle = 0.0f;
  } else {
    int best = 0;
    float bv = a_logit[0];
    for (int j = 1; j < nbin; ++j)
      if (a_logit[j] > bv) { bv = a_logit[j]; best = j; }  // first maximum, like torch.argmax
    d.bin = best;
    float ang = (2.0f * kPi / nbin) * best + a_res[best] * (kPi / nbin);
    if (ang > kPi) ang -= 2.0f * kPi;
    d.angle = ang;
  }
  d.ca = cosf(d.angle);
  d.sa = sinf(d.angle);
  return d;
}

__global__ __launch_bounds__(256) void box_decode_fwd_kernel(const float *__restrict__ center_raw, const float *__restrict__ size_raw,
                                                             const float *__restrict__ angle_logits,
                                                             const float *__restrict__ angle_res_norm,
                                                             const float *__restrict__ cls_logits, Strides st,
                                                             const float *__restrict__ query_xyz, const float *__restrict__ dims_min,
                                                             const float *__restrict__ dims_max, int nl, int b, int nq, int nbin,
                                                             int ncls1, float *center_norm, float *center_unnorm, float *size_norm,
                                                             float *size_unnorm, float *angle_residual, float *angle_cont, float *corners,
                                                             float *corners_xyz, float *cls_prob, float *obj_prob) {
  const long long row = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (row >= static_cast<long long>(nl) * b * nq) return;
  const int q = static_cast<int>(row % nq), bi = static_cast<int>((row / nq) % b), l = static_cast<int>(row / (static_cast<long long>(nq) * b));
  const float *c_raw = row_ptr(center_raw, st, 0, l, bi, q), *s_raw = row_ptr(size_raw, st, 1, l, bi, q);
  const float *a_logit = row_ptr(angle_logits, st, 2, l, bi, q), *a_res = row_ptr(angle_res_norm, st, 3, l, bi, q);
  const float *cls = row_ptr(cls_logits, st, 4, l, bi, q);
  const Decoded d = decode_row(c_raw, s_raw, a_logit, a_res, query_xyz + (static_cast<size_t>(bi) * nq + q) * 3,
                               dims_min + bi * 3, dims_max + bi * 3, nbin);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    center_unnorm[row * 3 + a] = d.cu[a];
    center_norm[row * 3 + a] = (d.cu[a] - dims_min[bi * 3 + a]) / (dims_max[bi * 3 + a] - dims_min[bi * 3 + a]);
    size_norm[row * 3 + a] = d.sig_s[a];
    size_unnorm[row * 3 + a] = d.su[a];
  }
  for (int j = 0; j < nbin; ++j) angle_residual[row * nbin + j] = a_res[j] * (kPi / nbin);
  angle_cont[row] = d.angle;
  // camera frame: centre (x, -z, y), half extents (l, h, w) / 2, rotation about y
  const float ccx = d.cu[0], ccy = -d.cu[2], ccz = d.cu[1];
  const float hl = 0.5f * d.su[0], hw = 0.5f * d.su[1], hh = 0.5f * d.su[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float x = kSignCam[k][0] * hl, y = kSignCam[k][1] * hh, z = kSignCam[k][2] * hw;
    float *o = corners + (row * 8 + k) * 3;
    o[0] = d.ca * x + d.sa * z + ccx;
    o[1] = y + ccy;
    o[2] = -d.sa * x + d.ca * z + ccz;
    const float xx = kSignXyz[k][0] * hl, yy = kSignXyz[k][1] * hw, zz = kSignXyz[k][2] * hh;
    float *ox = corners_xyz + (row * 8 + k) * 3;  // depth frame: rotation about z by -angle
    ox[0] = d.ca * xx + d.sa * yy + d.cu[0];
    ox[1] = -d.sa * xx + d.ca * yy + d.cu[1];
    ox[2] = zz + d.cu[2];
  }
  float mx = cls[0];
  for (int j = 1; j < ncls1; ++j) mx = fmaxf(mx, cls[j]);
  float sum = 0.0f;
  for (int j = 0; j < ncls1; ++j) sum += __expf(cls[j] - mx);
  const float inv = 1.0f / sum;
  for (int j = 0; j < ncls1 - 1; ++j) cls_prob[row * (ncls1 - 1) + j] = __expf(cls[j] - mx) * inv;
  obj_prob[row] = 1.0f - __expf(cls[ncls1 - 1] - mx) * inv;
}

__global__ __launch_bounds__(256) void box_decode_bwd_kernel(const float *__restrict__ center_raw, const float *__restrict__ size_raw,
                                                             const float *__restrict__ angle_logits,
                                                             const float *__restrict__ angle_res_norm, Strides st,
                                                             const float *__restrict__ dims_min, const float *__restrict__ dims_max,
                                                             int nl, int b, int nq, int nbin, const float *g_cn, const float *g_cu,
                                                             const float *g_sn, const float *g_su, const float *g_ar, const float *g_ang,
                                                             const float *g_cor, const float *g_cxyz, float *d_center_raw,
                                                             float *d_size_raw, float *d_angle_res_norm) {
  const long long row = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (row >= static_cast<long long>(nl) * b * nq) return;
  const int q = static_cast<int>(row % nq), bi = static_cast<int>((row / nq) % b), l = static_cast<int>(row / (static_cast<long long>(nq) * b));
  const float zero3[3] = {0.f, 0.f, 0.f};
  const Decoded d = decode_row(row_ptr(center_raw, st, 0, l, bi, q), row_ptr(size_raw, st, 1, l, bi, q),
                               row_ptr(angle_logits, st, 2, l, bi, q), row_ptr(angle_res_norm, st, 3, l, bi, q), zero3,
                               dims_min + bi * 3, dims_max + bi * 3, nbin);
  float gcu[3], gsu[3], ga = g_ang ? g_ang[row] : 0.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gcu[a] = (g_cu ? g_cu[row * 3 + a] : 0.0f) + (g_cn ? g_cn[row * 3 + a] * d.inv_range[a] : 0.0f);
    gsu[a] = g_su ? g_su[row * 3 + a] : 0.0f;
  }
  const float hl = 0.5f * d.su[0], hw = 0.5f * d.su[1];  // (the height only enters un-rotated: no term needs it)
  float ghl = 0.f, ghw = 0.f, ghh = 0.f;
  if (g_cor) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float *g = g_cor + (row * 8 + k) * 3;
      const float x = kSignCam[k][0] * hl, z = kSignCam[k][2] * hw;
      const float xr = d.ca * x + d.sa * z, zr = -d.sa * x + d.ca * z;  // rotated, before the centre is added
      gcu[0] += g[0]; gcu[2] -= g[1]; gcu[1] += g[2];                    // centre (x, -z, y)
      ghl += kSignCam[k][0] * (d.ca * g[0] - d.sa * g[2]);
      ghh += kSignCam[k][1] * g[1];
      ghw += kSignCam[k][2] * (d.sa * g[0] + d.ca * g[2]);
      ga += g[0] * zr - g[2] * xr;
    }
  }
  if (g_cxyz) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float *g = g_cxyz + (row * 8 + k) * 3;
      const float x = kSignXyz[k][0] * hl, y = kSignXyz[k][1] * hw;
      const float xr = d.ca * x + d.sa * y, yr = -d.sa * x + d.ca * y;
      gcu[0] += g[0]; gcu[1] += g[1]; gcu[2] += g[2];
      ghl += kSignXyz[k][0] * (d.ca * g[0] - d.sa * g[1]);
      ghw += kSignXyz[k][1] * (d.sa * g[0] + d.ca * g[1]);
      ghh += kSignXyz[k][2] * g[2];
      ga += g[0] * yr - g[1] * xr;
    }
  }
  gsu[0] += 0.5f * ghl; gsu[1] += 0.5f * ghw; gsu[2] += 0.5f * ghh;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    d_center_raw[row * 3 + a] = gcu[a] * d.sig_c[a] * (1.0f - d.sig_c[a]);
    const float gsn = (g_sn ? g_sn[row * 3 + a] : 0.0f) + gsu[a] * d.scale[a];
    d_size_raw[row * 3 + a] = gsn * d.sig_s[a] * (1.0f - d.sig_s[a]);
  }
  for (int j = 0; j < nbin; ++j) {
    float g = g_ar ? g_ar[row * nbin + j] : 0.0f;
    if (nbin > 1 && j == d.bin) g += ga;
    d_angle_res_norm[row * nbin + j] = g * (kPi / nbin);
  }
}

Strides make_strides(const long long *s) {
  Strides st;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 3; ++j) st.v[i][j] = s[i * 3 + j];
  return st;
}

}  // namespace
}  // namespace coda

CODA_API int coda_box_decode_fwd_f32(const float *center_raw, const float *size_raw, const float *angle_logits,
                                     const float *angle_res_norm, const float *cls_logits, const long long *strides,
                                     const float *query_xyz, const float *dims_min, const float *dims_max, int nl, int b,
                                     int nq, int nbin, int ncls1, float *center_norm, float *center_unnorm,
                                     float *size_norm, float *size_unnorm, float *angle_residual, float *angle_cont,
                                     float *corners, float *corners_xyz, float *cls_prob, float *obj_prob, void *stream) {
  using namespace coda;
  if (nl < 0 || b < 0 || nq < 0 || nbin < 1 || ncls1 < 2 || !strides) return CODA_EINVAL;
  const long long rows = static_cast<long long>(nl) * b * nq;
  if (rows == 0) return CODA_OK;
  if (!center_raw || !size_raw || !angle_logits || !angle_res_norm || !cls_logits || !query_xyz || !dims_min || !dims_max ||
      !center_norm || !center_unnorm || !size_norm || !size_unnorm || !angle_residual || !angle_cont || !corners ||
      !corners_xyz || !cls_prob || !obj_prob)
    return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(box_decode_fwd_kernel, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), center_raw, size_raw, angle_logits, angle_res_norm, cls_logits,
                     make_strides(strides), query_xyz, dims_min, dims_max, nl, b, nq, nbin, ncls1, center_norm, center_unnorm,
                     size_norm, size_unnorm, angle_residual, angle_cont, corners, corners_xyz, cls_prob, obj_prob);
  return launch_status();
}

CODA_API int coda_box_decode_bwd_f32(const float *center_raw, const float *size_raw, const float *angle_logits,
                                     const float *angle_res_norm, const long long *strides, const float *dims_min,
                                     const float *dims_max, int nl, int b, int nq, int nbin, const float *g_center_norm,
                                     const float *g_center_unnorm, const float *g_size_norm, const float *g_size_unnorm,
                                     const float *g_angle_residual, const float *g_angle_cont, const float *g_corners,
                                     const float *g_corners_xyz, float *d_center_raw, float *d_size_raw,
                                     float *d_angle_res_norm, void *stream) {
  using namespace coda;
  if (nl < 0 || b < 0 || nq < 0 || nbin < 1 || !strides) return CODA_EINVAL;
  const long long rows = static_cast<long long>(nl) * b * nq;
  if (rows == 0) return CODA_OK;
  if (!center_raw || !size_raw || !angle_logits || !angle_res_norm || !dims_min || !dims_max || !d_center_raw || !d_size_raw ||
      !d_angle_res_norm)
    return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(box_decode_bwd_kernel, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), center_raw, size_raw, angle_logits, angle_res_norm, make_strides(strides),
                     dims_min, dims_max, nl, b, nq, nbin, g_center_norm, g_center_unnorm, g_size_norm, g_size_unnorm,
                     g_angle_residual, g_angle_cont, g_corners, g_corners_xyz, d_center_raw, d_size_raw, d_angle_res_norm);
  return launch_status();
}
