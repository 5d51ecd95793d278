// eval_post.hip -- proposal filtering of the evaluation loop on the device.
//
// Replaces the per-proposal host loops of utils/ap_calculator.py (parse_predictions*, "remove_empty_box" and the
// utils/nms.py greedy suppression).  Two kernels:
//   box_point_count_kernel  b x ceil(k/8) workgroups; each streams the scene's points once and tests them against
//                           8 boxes held in registers (41 M point-in-box tests per 8 x 256 x 20 000 batch)
//   nms_kernel              one workgroup per scene: extents in fp64, bitonic sort of the candidates by score in
//                           LDS, then the greedy sweep -- the visit order is inherently serial, the suppression
//                           test of one kept box against all later candidates is the parallel part
#include "coda_eval.h"
#include "common.hip.h"

// Every expression below is evaluated as written, one IEEE operation per source operation (no fused multiply-add):
// the NMS overlaps are numpy's float64 expressions, and the point-in-box test then agrees bit for bit with its
// restatement in oracle/eval_oracle.py.
#pragma clang fp contract(off)

namespace coda {
namespace {

constexpr int kCountThreads = 256;
constexpr int kBoxesPerWg = 8;

struct Frame {  // origin + three edges with their squared lengths
  float o[3], e[3][3], len2[3];
};

__global__ __launch_bounds__(kCountThreads) void box_point_count_kernel(const float *__restrict__ corners,
                                                                        const float *__restrict__ points,
                                                                        int32_t *__restrict__ counts, int k, int n,
                                                                        int stride, int nscenes) {
  const int scene = blockIdx.x % nscenes, chunk = blockIdx.x / nscenes;  // one scene's workgroups share an XCD's L2
  const int first = chunk * kBoxesPerWg;
  __shared__ Frame s_frame[kBoxesPerWg];
  __shared__ int s_count[kBoxesPerWg];
  if (threadIdx.x < kBoxesPerWg) {
    s_count[threadIdx.x] = 0;
    const int box = min(first + static_cast<int>(threadIdx.x), k - 1);
    const float *c = corners + (static_cast<size_t>(scene) * k + box) * 24;
    Frame f;
    const int other[3] = {1, 3, 4};
#pragma unroll
    for (int a = 0; a < 3; ++a) f.o[a] = c[a];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      float l2 = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        f.e[e][a] = c[other[e] * 3 + a] - c[a];
        l2 = l2 + f.e[e][a] * f.e[e][a];
      }
      f.len2[e] = l2 > 0.f ? l2 : -1.f;  // a box without extent along an edge holds nothing (t <= -1 never holds)
    }
    s_frame[threadIdx.x] = f;
  }
  __syncthreads();
  Frame f[kBoxesPerWg];
#pragma unroll
  for (int q = 0; q < kBoxesPerWg; ++q) f[q] = s_frame[q];
  int local[kBoxesPerWg] = {};
  const float *p = points + static_cast<size_t>(scene) * n * stride;
  for (int i = threadIdx.x; i < n; i += kCountThreads) {
    const float dx = p[static_cast<size_t>(i) * stride], dy = p[static_cast<size_t>(i) * stride + 1],
                dz = p[static_cast<size_t>(i) * stride + 2];
    const float cam[3] = {dx, -dz, dy};  // depth -> upright camera (flip_axis_to_camera, utils/box_util.py:64-70)
#pragma unroll
    for (int q = 0; q < kBoxesPerWg; ++q) {
      bool in = true;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const float t = (cam[0] - f[q].o[0]) * f[q].e[e][0] + (cam[1] - f[q].o[1]) * f[q].e[e][1] +
                        (cam[2] - f[q].o[2]) * f[q].e[e][2];
        in = in && t >= 0.f && t <= f[q].len2[e];
      }
      local[q] += in ? 1 : 0;
    }
  }
#pragma unroll
  for (int q = 0; q < kBoxesPerWg; ++q) {
    int v = local[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane_id() == 0 && v) atomicAdd(&s_count[q], v);
  }
  __syncthreads();
  if (threadIdx.x < kBoxesPerWg && first + static_cast<int>(threadIdx.x) < k)
    counts[static_cast<size_t>(scene) * k + first + threadIdx.x] = s_count[threadIdx.x];
}

constexpr int kNmsThreads = 256;
constexpr int kNmsMax = 2048;

struct Extent {
  double lo[3], hi[3], vol;
};

// numpy's expressions: utils/nms.py:98-116 / :59-73 / :146-168
__device__ __forceinline__ double overlap(const Extent &a, const Extent &b, int mode, int old_type) {
  double inter;
  if (mode == 0) {  // 2-D: x and z extents
    const double w = fmax(0.0, fmin(a.hi[0], b.hi[0]) - fmax(a.lo[0], b.lo[0]));
    const double h = fmax(0.0, fmin(a.hi[2], b.hi[2]) - fmax(a.lo[2], b.lo[2]));
    inter = w * h;
  } else {
    const double l = fmax(0.0, fmin(a.hi[0], b.hi[0]) - fmax(a.lo[0], b.lo[0]));
    const double w = fmax(0.0, fmin(a.hi[1], b.hi[1]) - fmax(a.lo[1], b.lo[1]));
    const double h = fmax(0.0, fmin(a.hi[2], b.hi[2]) - fmax(a.lo[2], b.lo[2]));
    inter = l * w * h;
  }
  if (old_type) return inter / b.vol;
  return inter / (a.vol + b.vol - inter);
}

__device__ __forceinline__ double volume(const Extent &e, int mode) {
  if (mode == 0) return (e.hi[0] - e.lo[0]) * (e.hi[2] - e.lo[2]);
  return (e.hi[0] - e.lo[0]) * (e.hi[1] - e.lo[1]) * (e.hi[2] - e.lo[2]);
}

__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const float *__restrict__ corners, const float *__restrict__ scores,
                                                          const int32_t *__restrict__ classes,
                                                          const unsigned char *__restrict__ nonempty,
                                                          unsigned char *__restrict__ keep, int k, int kpow2, int mode,
                                                          double nms_iou, int old_type) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Extent *s_ext = reinterpret_cast<Extent *>(smem);                       // [k]
  unsigned long long *s_key = reinterpret_cast<unsigned long long *>(s_ext + k);  // [kpow2] sort keys
  unsigned char *s_alive = reinterpret_cast<unsigned char *>(s_key + kpow2);      // [kpow2], by sorted position
  __shared__ int s_any, s_best;
  __shared__ unsigned long long s_bestkey;
  const int scene = blockIdx.x, tid = threadIdx.x;
  const float *c = corners + static_cast<size_t>(scene) * k * 24;
  const float *sc = scores + static_cast<size_t>(scene) * k;
  const int32_t *cls = classes ? classes + static_cast<size_t>(scene) * k : nullptr;
  const unsigned char *ne = nonempty ? nonempty + static_cast<size_t>(scene) * k : nullptr;
  if (tid == 0) { s_any = 0; s_bestkey = 0; s_best = 0; }
  __syncthreads();
  // descending (score, index) as one unsigned key: order-preserving map of the float bits, index in the low bits
  auto key_of = [&](int j) {
    unsigned u = __float_as_uint(sc[j]);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (static_cast<unsigned long long>(u) << 32) | static_cast<unsigned>(j);
  };
  for (int j = tid; j < k; j += kNmsThreads) {
    Extent e;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float lo = c[j * 24 + a], hi = lo;
#pragma unroll
      for (int q = 1; q < 8; ++q) {
        lo = fminf(lo, c[j * 24 + q * 3 + a]);
        hi = fmaxf(hi, c[j * 24 + q * 3 + a]);
      }
      e.lo[a] = lo;
      e.hi[a] = hi;
    }
    e.vol = volume(e, mode);
    s_ext[j] = e;
    if (!ne || ne[j]) s_any = 1;
    keep[static_cast<size_t>(scene) * k + j] = 0;
  }
  __syncthreads();
  if (!s_any) {  // no box holds points: the most object-like box stands in (np.argmax: first maximum)
    for (int j = tid; j < k; j += kNmsThreads) {
      unsigned u = __float_as_uint(sc[j]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      atomicMax(&s_bestkey, (static_cast<unsigned long long>(u) << 32) | static_cast<unsigned>(k - 1 - j));
    }
    __syncthreads();
    if (tid == 0) s_best = k - 1 - static_cast<int>(s_bestkey & 0xffffffffu);
    __syncthreads();
  }
  const bool any = s_any != 0;
  const int best = s_best;
  for (int j = tid; j < kpow2; j += kNmsThreads) {
    const bool cand = j < k && (any ? (!ne || ne[j]) : j == best);
    s_key[j] = cand ? key_of(j) : 0ull;  // non-candidates sort to the end (real keys are > 0)
  }
  __syncthreads();
  for (int size = 2; size <= kpow2; size <<= 1) {  // bitonic sort, descending
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      for (int t = tid; t < kpow2 / 2; t += kNmsThreads) {
        const int lo = 2 * t - (t & (strd - 1)), hi = lo + strd;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = s_key[lo], b = s_key[hi];
        if ((a < b) == desc) {
          s_key[lo] = b;
          s_key[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int j = tid; j < kpow2; j += kNmsThreads) s_alive[j] = s_key[j] != 0ull;
  __syncthreads();
  for (int i = 0; i < k; ++i) {
    if (s_key[i] == 0ull) break;  // uniform: past the last candidate
    if (!s_alive[i]) continue;    // uniform
    const int bi = static_cast<int>(s_key[i] & 0xffffffffu);
    if (tid == 0) keep[static_cast<size_t>(scene) * k + bi] = 1;
    const Extent ei = s_ext[bi];
    const int ci = (mode == 2) ? cls[bi] : 0;
    for (int j = i + 1 + tid; j < k; j += kNmsThreads) {
      if (!s_alive[j]) continue;
      const int bj = static_cast<int>(s_key[j] & 0xffffffffu);
      double o = overlap(ei, s_ext[bj], mode, old_type);
      if (mode == 2 && cls[bj] != ci) o = o * 0.0;  // o * (cls1 == cls2): NaN stays NaN, as in numpy
      if (o > nms_iou) s_alive[j] = 0;
    }
    __syncthreads();
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_box_point_count_f32(const float *corners, const float *points, int32_t *counts, int b, int k, int n,
                                      int point_stride, void *stream) {
  using namespace coda;
  if (b < 0 || k < 0 || n < 0 || point_stride < 3) return CODA_EINVAL;
  if (b == 0 || k == 0) return CODA_OK;
  if (!corners || !counts || (n > 0 && !points)) return CODA_EINVAL;
  clear_sticky_error();
  const int chunks = (k + kBoxesPerWg - 1) / kBoxesPerWg;
  hipLaunchKernelGGL(box_point_count_kernel, dim3(static_cast<unsigned>(b) * chunks), dim3(kCountThreads), 0,
                     static_cast<hipStream_t>(stream), corners, points, counts, k, n, point_stride, b);
  return launch_status();
}

CODA_API int coda_nms_f32(const float *corners, const float *scores, const int32_t *classes, const unsigned char *nonempty,
                          unsigned char *keep, int b, int k, int mode, double nms_iou, int old_type, void *stream) {
  using namespace coda;
  if (b < 0 || k < 0 || mode < 0 || mode > 2) return CODA_EINVAL;
  if (b == 0 || k == 0) return CODA_OK;
  if (!corners || !scores || !keep || (mode == 2 && !classes)) return CODA_EINVAL;
  if (k > kNmsMax) return CODA_ENOSPC;
  int kpow2 = 2;
  while (kpow2 < k) kpow2 <<= 1;
  const size_t lds = sizeof(Extent) * k + sizeof(unsigned long long) * kpow2 + kpow2;
  auto kern = nms_kernel;
  if (int st = raise_dynamic_lds(kern, lds); st != CODA_OK) return st;
  clear_sticky_error();
  hipLaunchKernelGGL(kern, dim3(b), dim3(kNmsThreads), lds, static_cast<hipStream_t>(stream), corners, scores, classes,
                     nonempty, keep, k, kpow2, mode, nms_iou, old_type);
  return launch_status();
}
