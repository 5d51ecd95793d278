// box_loss.hip -- the matched box losses of SetCriterion for all decoder layers in one kernel per direction.
//
// Replaces loss_sem_cls_softmax_skip_none_gt_sample (criterion.py:219-246), loss_angle (:834-900),
// loss_center (:1015-1039) and loss_size (:1065-1104) as evaluated per decoder layer by
// single_output_forward (:1106-1160): in PyTorch ~60 gather / where / cross-entropy / reduction launches of
// 3-10 us per direction on 16 384 proposal rows, issued slower than the GPU retires them.  One thread per
// proposal row; per-row partials out, the caller reduces per layer and applies the reference's normalisers.
#include "coda_box_ops.h"
#include "common.hip.h"

namespace coda {
namespace {

struct LossStrides {
  long long v[5][3];
};
struct LossArgs {
  const float *sem, *ang, *res, *cen, *siz;
  LossStrides st;
  const int64_t *gt_inds;
  const float *matched;
  const int64_t *gt_sem, *gt_ang;
  const float *gt_res, *gt_cen, *gt_siz, *has_obj, *sem_w;
  int nl, b, nq, ngt, nsem, nbin;
};

__device__ __forceinline__ const float *rp(const float *base, const LossStrides &st, int which, int l, int bi, int q) {
  return base + l * st.v[which][0] + bi * st.v[which][1] + q * st.v[which][2];
}

// log-sum-exp of n logits
__device__ __forceinline__ float lse(const float *x, int n) {
  float mx = x[0];
  for (int j = 1; j < n; ++j) mx = fmaxf(mx, x[j]);
  float s = 0.f;
  for (int j = 0; j < n; ++j) s += __expf(x[j] - mx);
  return mx + __logf(s);
}

template <bool BWD>
__global__ __launch_bounds__(256) void box_loss_kernel(LossArgs a, float *__restrict__ partial, const float *__restrict__ g,
                                                       float *d_sem, float *d_ang, float *d_res, float *d_cen, float *d_siz) {
  const long long row = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (row >= static_cast<long long>(a.nl) * a.b * a.nq) return;
  const int q = static_cast<int>(row % a.nq), bi = static_cast<int>((row / a.nq) % a.b), l = static_cast<int>(row / (static_cast<long long>(a.nq) * a.b));
  const float *sem = rp(a.sem, a.st, 0, l, bi, q), *ang = rp(a.ang, a.st, 1, l, bi, q), *res = rp(a.res, a.st, 2, l, bi, q);
  const float *cen = rp(a.cen, a.st, 3, l, bi, q), *siz = rp(a.siz, a.st, 4, l, bi, q);
  const int gi = static_cast<int>(a.gt_inds[row]);
  const float m = a.matched[row];
  const bool is_matched = static_cast<int>(m) != 0;  // `proposal_matched_mask.int() == 0` (criterion.py:226)
  const size_t gtrow = static_cast<size_t>(bi) * a.ngt + gi;
  // semantic class: matched -> GT label, else background
  const int y = is_matched ? static_cast<int>(a.gt_sem[gtrow]) : a.nsem - 1;
  const float wy = a.sem_w[y] * a.has_obj[bi];
  const float sem_lse = lse(sem, a.nsem);
  const int yb = static_cast<int>(a.gt_ang[gtrow]);
  const float ang_lse = lse(ang, a.nbin);
  const float err = res[yb] - a.gt_res[gtrow];
  const float abs_err = fabsf(err), quad = fminf(abs_err, 1.0f);
  if (!BWD) {
    float *p = partial + row * 5;
    p[0] = wy * (sem_lse - sem[y]);
    p[1] = m * (ang_lse - ang[yb]);
    p[2] = m * (0.5f * quad * quad + (abs_err - quad));  // huber, delta = 1 (utils/misc.py)
    float dc = 0.f, ds = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dc += fabsf(cen[c] - a.gt_cen[gtrow * 3 + c]);
      ds += fabsf(siz[c] - a.gt_siz[gtrow * 3 + c]);
    }
    p[3] = m * dc;
    p[4] = m * ds;
  } else {
    const float *gl = g + l * 5;
    for (int j = 0; j < a.nsem; ++j)
      d_sem[row * a.nsem + j] = gl[0] * wy * (__expf(sem[j] - sem_lse) - (j == y ? 1.f : 0.f));
    const float derr = abs_err <= 1.0f ? err : (err > 0.f ? 1.f : -1.f);
    for (int j = 0; j < a.nbin; ++j) {
      d_ang[row * a.nbin + j] = gl[1] * m * (__expf(ang[j] - ang_lse) - (j == yb ? 1.f : 0.f));
      d_res[row * a.nbin + j] = j == yb ? gl[2] * m * derr : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ec = cen[c] - a.gt_cen[gtrow * 3 + c], es = siz[c] - a.gt_siz[gtrow * 3 + c];
      d_cen[row * 3 + c] = gl[3] * m * (ec > 0.f ? 1.f : (ec < 0.f ? -1.f : 0.f));
      d_siz[row * 3 + c] = gl[4] * m * (es > 0.f ? 1.f : (es < 0.f ? -1.f : 0.f));
    }
  }
}

int fill(LossArgs &a, const float *sem_logits, const float *angle_logits, const float *angle_res_norm, const float *center_norm,
         const float *size_norm, const long long *strides, const int64_t *gt_inds, const float *matched,
         const int64_t *gt_sem_label, const int64_t *gt_angle_class, const float *gt_res_norm, const float *gt_center,
         const float *gt_size, const float *has_object, const float *sem_class_weight, int nl, int b, int nq, int ngt, int nsem,
         int nbin) {
  if (nl < 0 || b < 0 || nq < 0 || ngt <= 0 || nsem < 1 || nbin < 1 || nsem > 64 || nbin > 64 || !strides) return CODA_EINVAL;
  if (!sem_logits || !angle_logits || !angle_res_norm || !center_norm || !size_norm || !gt_inds || !matched || !gt_sem_label ||
      !gt_angle_class || !gt_res_norm || !gt_center || !gt_size || !has_object || !sem_class_weight)
    return CODA_EINVAL;
  a.sem = sem_logits; a.ang = angle_logits; a.res = angle_res_norm; a.cen = center_norm; a.siz = size_norm;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 3; ++j) a.st.v[i][j] = strides[i * 3 + j];
  a.gt_inds = gt_inds; a.matched = matched; a.gt_sem = gt_sem_label; a.gt_ang = gt_angle_class; a.gt_res = gt_res_norm;
  a.gt_cen = gt_center; a.gt_siz = gt_size; a.has_obj = has_object; a.sem_w = sem_class_weight;
  a.nl = nl; a.b = b; a.nq = nq; a.ngt = ngt; a.nsem = nsem; a.nbin = nbin;
  return CODA_OK;
}

}  // namespace
}  // namespace coda

CODA_API int coda_box_loss_fwd_f32(const float *sem_logits, const float *angle_logits, const float *angle_res_norm,
                                   const float *center_norm, const float *size_norm, const long long *strides,
                                   const int64_t *gt_inds, const float *matched, const int64_t *gt_sem_label,
                                   const int64_t *gt_angle_class, const float *gt_res_norm, const float *gt_center,
                                   const float *gt_size, const float *has_object, const float *sem_class_weight, int nl, int b,
                                   int nq, int ngt, int nsem, int nbin, float *partial, void *stream) {
  using namespace coda;
  const long long rows = static_cast<long long>(nl) * b * nq;
  if (rows == 0) return CODA_OK;
  LossArgs a;
  const int st = fill(a, sem_logits, angle_logits, angle_res_norm, center_norm, size_norm, strides, gt_inds, matched, gt_sem_label,
                      gt_angle_class, gt_res_norm, gt_center, gt_size, has_object, sem_class_weight, nl, b, nq, ngt, nsem, nbin);
  if (st != CODA_OK || !partial) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(box_loss_kernel<false>, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, partial, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return launch_status();
}

CODA_API int coda_box_loss_bwd_f32(const float *sem_logits, const float *angle_logits, const float *angle_res_norm,
                                   const float *center_norm, const float *size_norm, const long long *strides,
                                   const int64_t *gt_inds, const float *matched, const int64_t *gt_sem_label,
                                   const int64_t *gt_angle_class, const float *gt_res_norm, const float *gt_center,
                                   const float *gt_size, const float *has_object, const float *sem_class_weight, int nl, int b,
                                   int nq, int ngt, int nsem, int nbin, const float *g, float *d_sem, float *d_angle, float *d_res,
                                   float *d_center, float *d_size, void *stream) {
  using namespace coda;
  const long long rows = static_cast<long long>(nl) * b * nq;
  if (rows == 0) return CODA_OK;
  LossArgs a;
  const int st = fill(a, sem_logits, angle_logits, angle_res_norm, center_norm, size_norm, strides, gt_inds, matched, gt_sem_label,
                      gt_angle_class, gt_res_norm, gt_center, gt_size, has_object, sem_class_weight, nl, b, nq, ngt, nsem, nbin);
  if (st != CODA_OK || !g || !d_sem || !d_angle || !d_res || !d_center || !d_size) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(box_loss_kernel<true>, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, nullptr, g, d_sem, d_angle, d_res, d_center, d_size);
  return launch_status();
}
