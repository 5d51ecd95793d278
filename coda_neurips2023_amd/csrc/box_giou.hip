// box_giou.hip -- generalized 3D IoU between every proposal and every GT box of a scene.
//
// Replaces utils/box_util.py:655-745 (generalized_box3d_iou_tensor) incl. the Sutherland-Hodgman
// polygon clip (:526-577) that the reference runs on the host in a Python triple loop.  One thread
// per (scene, proposal, GT box) pair; the two quads and the <= 8-vertex clip polygon live in
// registers / a small per-thread array.  fp32 throughout, operations in the reference's order.
#include "coda_box_ops.h"
#include "common.hip.h"

namespace coda {
namespace {

struct P2 {
  float x, y;
};

// helper_inside (:520-523): strictly left of the directed clip edge cp1 -> cp2
__device__ __forceinline__ bool inside(P2 cp1, P2 cp2, P2 p) {
  return __fmul_rn(cp2.x - cp1.x, p.y - cp1.y) > __fmul_rn(cp2.y - cp1.y, p.x - cp1.x);
}
// helper_computeIntersection (:509-517)
__device__ __forceinline__ P2 intersect(P2 cp1, P2 cp2, P2 s, P2 e) {
  const float dcx = cp1.x - cp2.x, dcy = cp1.y - cp2.y;
  const float dpx = s.x - e.x, dpy = s.y - e.y;
  const float n1 = __fsub_rn(__fmul_rn(cp1.x, cp2.y), __fmul_rn(cp1.y, cp2.x));
  const float n2 = __fsub_rn(__fmul_rn(s.x, e.y), __fmul_rn(s.y, e.x));
  const float n3 = 1.0f / __fsub_rn(__fmul_rn(dcx, dpy), __fmul_rn(dcy, dpx));
  return {__fmul_rn(__fsub_rn(__fmul_rn(n1, dpx), __fmul_rn(n2, dcx)), n3),
          __fmul_rn(__fsub_rn(__fmul_rn(n1, dpy), __fmul_rn(n2, dcy)), n3)};
}

// area of clip(subject, clipper): polygon_clip_unnest (:526-577) + the shoelace sum (:723-727)
__device__ float clipped_area(const P2 (&subject)[4], const P2 (&clipper)[4]) {
  P2 out[10], in[10];
  int nout = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = subject[i];
  P2 cp1 = clipper[3];
  for (int c = 0; c < 4; ++c) {
    const P2 cp2 = clipper[c];
    const int nin = nout;
    for (int i = 0; i < nin; ++i) in[i] = out[i];
    nout = 0;
    P2 s = in[nin - 1];
    for (int i = 0; i < nin; ++i) {
      const P2 e = in[i];
      if (inside(cp1, cp2, e)) {
        if (!inside(cp1, cp2, s)) out[nout++] = intersect(cp1, cp2, s, e);
        out[nout++] = e;
      } else if (inside(cp1, cp2, s)) {
        out[nout++] = intersect(cp1, cp2, s, e);
      }
      s = e;
    }
    cp1 = cp2;
    if (nout == 0) return 0.0f;
  }
  // | dot(xs, roll(ys, 1)) - dot(ys, roll(xs, 1)) | * 0.5
  float a = 0.0f, b = 0.0f;
  for (int i = 0; i < nout; ++i) {
    const P2 prev = out[(i + nout - 1) % nout];
    a += out[i].x * prev.y;
    b += out[i].y * prev.x;
  }
  return fabsf(a - b) * 0.5f;
}

__device__ __forceinline__ float edge_len(const float *c, int i, int j) {  // box3d_vol_tensor (:580-600)
  const float dx = c[i * 3] - c[j * 3], dy = c[i * 3 + 1] - c[j * 3 + 1], dz = c[i * 3 + 2] - c[j * 3 + 2];
  return sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-6f));
}

// The matcher's cost matrix next to the gIoU (criterion.py:58-66): -prob[label] * w_class - objectness *
// w_objectness + L1(centres) * w_center - gIoU * w_giou, one rounding per operation in the reference's order
// (explicit _rn intrinsics: no FMA contraction), so the device costs are the ones the torch expressions give.
struct CostArgs {
  const float *center1, *center2, *cls_prob, *objectness;
  const int64_t *labels;
  float *center_dist, *cost;
  float w_class, w_objectness, w_center, w_giou;
  int ncls;
};

__global__ __launch_bounds__(256) void giou_kernel(const float *__restrict__ corners1, const float *__restrict__ corners2,
                                                   const int32_t *__restrict__ nums_k2, float *__restrict__ out,
                                                   int k1n, int k2n, long long total, int rotated, int vols_only,
                                                   int k2_limit, const unsigned char *__restrict__ rotated_dev,
                                                   CostArgs ca) {
  const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= total) return;
  if (rotated_dev) rotated = *rotated_dev != 0;  // the flag lives on the device: no host read-back of the angles
  const int k2 = static_cast<int>(t % k2n), k1 = static_cast<int>((t / k2n) % k1n);
  const int b = static_cast<int>(t / (static_cast<long long>(k2n) * k1n));
  const bool real = !nums_k2 || k2 < nums_k2[b];
  float c1[24], c2[24];
  const float *p1 = corners1 + (static_cast<size_t>(b) * k1n + k1) * 24;
  const float *p2 = corners2 + (static_cast<size_t>(b) * k2n + k2) * 24;
#pragma unroll
  for (int i = 0; i < 24; ++i) { c1[i] = p1[i]; c2[i] = p2[i]; }

  // height (Y is negative-up): :676-678
  const float ymax = fminf(c1[0 * 3 + 1], c2[0 * 3 + 1]), ymin = fmaxf(c1[4 * 3 + 1], c2[4 * 3 + 1]);
  const float height = fmaxf(ymax - ymin, 0.0f);
  // rect = corners (3, 2, 1, 0) x components (x, z): :681-686
  P2 r1[4], r2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r1[i] = {c1[(3 - i) * 3], c1[(3 - i) * 3 + 2]};
    r2[i] = {c2[(3 - i) * 3], c2[(3 - i) * 3 + 2]};
  }
  const float ltx = fmaxf(r1[1].x, r2[1].x), lty = fmaxf(r1[1].y, r2[1].y);
  const float rbx = fminf(r1[3].x, r2[3].x), rby = fminf(r1[3].y, r2[3].y);
  float area = fmaxf(rbx - ltx, 0.0f) * fmaxf(rby - lty, 0.0f);
  if (!real) area = 0.0f;  // :692-694
  if (rotated) {
    // vols_only == 2: the evaluation's box3d_iou (utils/box_util.py:156-183) clips EVERY pair; the matcher's
    // generalized_box3d_iou_tensor first tests corners 1 / 3 of the two quadrilaterals as if they were axis-aligned
    // extents (:688-694) and skips the pairs that test rejects -- reproduced for the matcher, not wanted here
    const bool visit = real && (vols_only == 2 || (area != 0.0f && (k2_limit < 0 || k2 < k2_limit)));
    area = visit ? clipped_area(r1, r2) : 0.0f;
  }
  const float inter_vol = area * height;
  if (vols_only) {
    out[t] = inter_vol;
    return;
  }
  // enclosing axis-aligned box with Y flipped (:603-653)
  float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
  float y1max = -INFINITY, y1min = INFINITY, y2max = -INFINITY, y2min = INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xmin = fminf(xmin, fminf(c1[i * 3], c2[i * 3]));
    xmax = fmaxf(xmax, fmaxf(c1[i * 3], c2[i * 3]));
    zmin = fminf(zmin, fminf(c1[i * 3 + 2], c2[i * 3 + 2]));
    zmax = fmaxf(zmax, fmaxf(c1[i * 3 + 2], c2[i * 3 + 2]));
    y1max = fmaxf(y1max, -c1[i * 3 + 1]); y1min = fminf(y1min, -c1[i * 3 + 1]);
    y2max = fmaxf(y2max, -c2[i * 3 + 1]); y2min = fminf(y2min, -c2[i * 3 + 1]);
  }
  const float al_ymin = fmaxf(y1max, y2max), al_ymax = fminf(y1min, y2min);  // (sic) :622-637
  const float enclosing = fabsf(xmax - xmin) * fabsf(al_ymax - al_ymin) * fabsf(zmax - zmin);
  const float eps = 1e-8f;
  const float vol1 = fmaxf(edge_len(c1, 0, 1) * edge_len(c1, 1, 2) * edge_len(c1, 0, 4), eps);
  const float vol2 = fmaxf(edge_len(c2, 0, 1) * edge_len(c2, 1, 2) * edge_len(c2, 0, 4), eps);
  const float sum_vols = vol1 + vol2;
  const bool good = enclosing > 2 * eps && sum_vols > 4 * eps;
  const float union_vol = fmaxf(sum_vols - inter_vol, eps);
  float giou = inter_vol / union_vol - (1.0f - union_vol / enclosing);
  if (!good || !real) giou = 0.0f;
  out[t] = giou;
  if (ca.cost) {
    const float *a = ca.center1 + (static_cast<size_t>(b) * k1n + k1) * 3;
    const float *g = ca.center2 + (static_cast<size_t>(b) * k2n + k2) * 3;
    // torch.cdist(p=1) on 3 coordinates: lanes 0..2 of a shuffle-down tree, i.e. (|d0| + |d2|) + |d1|
    const float d0 = fabsf(__fsub_rn(a[0], g[0])), d1 = fabsf(__fsub_rn(a[1], g[1])), d2 = fabsf(__fsub_rn(a[2], g[2]));
    const float l1 = __fadd_rn(__fadd_rn(d0, d2), d1);
    ca.center_dist[t] = l1;
    const int64_t label = ca.labels[static_cast<size_t>(b) * k2n + k2];
    const float p = ca.cls_prob[(static_cast<size_t>(b) * k1n + k1) * ca.ncls + label];
    const float o = ca.objectness[static_cast<size_t>(b) * k1n + k1];
    float c = __fadd_rn(__fmul_rn(ca.w_class, -p), __fmul_rn(ca.w_objectness, -o));
    c = __fadd_rn(c, __fmul_rn(ca.w_center, l1));
    c = __fadd_rn(c, __fmul_rn(ca.w_giou, -giou));
    ca.cost[t] = c;
  }
}

}  // namespace
}  // namespace coda

namespace {
int launch_giou(const float *corners1, const float *corners2, const int32_t *nums_k2, float *out, int b, int k1, int k2,
                int rotated, const unsigned char *rotated_dev, int inter_vols_only, int rotated_k2_limit, void *stream,
                coda::CostArgs ca = coda::CostArgs{});
}

CODA_API int coda_generalized_box3d_iou_f32(const float *corners1, const float *corners2, const int32_t *nums_k2,
                                            float *out, int b, int k1, int k2, int rotated, int inter_vols_only,
                                            int rotated_k2_limit, void *stream) {
  return launch_giou(corners1, corners2, nums_k2, out, b, k1, k2, rotated, nullptr, inter_vols_only, rotated_k2_limit,
                     stream);
}

CODA_API int coda_generalized_box3d_iou_devflag_f32(const float *corners1, const float *corners2, const int32_t *nums_k2,
                                                    float *out, int b, int k1, int k2, const unsigned char *rotated_flag,
                                                    int inter_vols_only, int rotated_k2_limit, void *stream) {
  if (!rotated_flag) return CODA_EINVAL;
  return launch_giou(corners1, corners2, nums_k2, out, b, k1, k2, 0, rotated_flag, inter_vols_only, rotated_k2_limit,
                     stream);
}

CODA_API int coda_matcher_cost_f32(const float *corners1, const float *corners2, const int32_t *nums_k2,
                                   const float *center1, const float *center2, const float *cls_prob,
                                   const int64_t *labels, const float *objectness, float w_class, float w_objectness,
                                   float w_center, float w_giou, float *gious, float *center_dist, float *cost, int b,
                                   int k1, int k2, int ncls, int rotated, const unsigned char *rotated_flag,
                                   int rotated_k2_limit, void *stream) {
  if (ncls <= 0) return CODA_EINVAL;
  if (static_cast<long long>(b) * k1 * k2 > 0 &&
      (!center1 || !center2 || !cls_prob || !labels || !objectness || !center_dist || !cost))
    return CODA_EINVAL;
  coda::CostArgs ca{center1, center2, cls_prob, objectness, labels, center_dist, cost,
                    w_class, w_objectness, w_center, w_giou, ncls};
  return launch_giou(corners1, corners2, nums_k2, gious, b, k1, k2, rotated, rotated_flag, 0, rotated_k2_limit, stream, ca);
}

namespace {
int launch_giou(const float *corners1, const float *corners2, const int32_t *nums_k2, float *out, int b, int k1, int k2,
                int rotated, const unsigned char *rotated_dev, int inter_vols_only, int rotated_k2_limit, void *stream,
                coda::CostArgs ca) {
  using namespace coda;
  if (b < 0 || k1 < 0 || k2 < 0) return CODA_EINVAL;
  const long long total = static_cast<long long>(b) * k1 * k2;
  if (total == 0) return CODA_OK;
  if (!corners1 || !corners2 || !out) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(giou_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), corners1, corners2, nums_k2, out, k1, k2, total, rotated,
                     inter_vols_only, rotated_k2_limit, rotated_dev, ca);
  return launch_status();
}
}  // namespace
