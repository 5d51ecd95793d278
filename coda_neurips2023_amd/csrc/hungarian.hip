// hungarian.hip -- the matcher's assignment problem on the device (second half).
//
// Replaces scipy.optimize.linear_sum_assignment as Matcher.forward calls it per scene (criterion.py:68-79) --
// here once per (decoder layer, scene) problem, all problems in one launch, no host round trip.  Algorithm:
// shortest augmenting paths on the transposed problem (rows = the n real GT boxes, columns = the m >= n
// proposals), the formulation of scipy's rectangular LSAP solver: for every GT row, grow a tree of alternating
// paths with Dijkstra-like reduced costs until an unassigned proposal is reached, update the dual variables,
// flip the path.  One workgroup per problem: the columns live across the threads (reduced-cost update and
// arg-min are wave reductions), the row loop is serial (n <= 128 augmentations of at most n steps each).
// Costs are fp32 (exactly representable in the fp64 arithmetic of the duals, as in scipy).
#include "coda_box_ops.h"
#include "common.hip.h"

#include <cfloat>

namespace coda {
namespace {

constexpr int kHungThreads = 256;
constexpr int kMaxRows = 128;

struct Best {  // arg-min key: (value, column already assigned?, column index)
  double val;
  int taken, col;
};
__device__ __forceinline__ bool better(const Best &a, const Best &b) {
  if (a.val != b.val) return a.val < b.val;
  if (a.taken != b.taken) return a.taken < b.taken;  // prefer a column that ends the search (scipy does too)
  return a.col < b.col;
}

__global__ __launch_bounds__(kHungThreads) void hungarian_kernel(const float *__restrict__ cost, const int64_t *__restrict__ nactual,
                                                                 int64_t *__restrict__ out_inds, float *__restrict__ out_mask,
                                                                 int m, int ngt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // layout: v[m] f64 | spc[m] f64 | path[m] i32 | row4col[m] i32 | insc[m] i32 | cost^T[n][m] f32
  double *s_v = reinterpret_cast<double *>(smem);
  double *s_spc = s_v + m;
  int *s_path = reinterpret_cast<int *>(s_spc + m);
  int *s_row4col = s_path + m;
  int *s_insc = s_row4col + m;
  float *s_cost = reinterpret_cast<float *>(s_insc + m);
  __shared__ double s_u[kMaxRows];
  __shared__ int s_col4row[kMaxRows];
  __shared__ int s_insr[kMaxRows];
  __shared__ Best s_best[kHungThreads / kWave];
  __shared__ Best s_pick;
  __shared__ int s_invalid;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int prob = blockIdx.x;
  const int n = static_cast<int>(min(static_cast<int64_t>(ngt), max(static_cast<int64_t>(0), nactual[prob])));
  const float *c = cost + static_cast<size_t>(prob) * m * ngt;
  int64_t *inds = out_inds + static_cast<size_t>(prob) * m;
  float *mask = out_mask + static_cast<size_t>(prob) * m;
  for (int j = tid; j < m; j += kHungThreads) {
    inds[j] = 0;
    mask[j] = 0.f;
    s_v[j] = 0.0;
    s_row4col[j] = -1;
  }
  if (n == 0) return;  // block-uniform
  if (tid == 0) s_invalid = 0;
  __syncthreads();
  bool bad = false;
  for (int e = tid; e < m * n; e += kHungThreads) {  // cost^T into LDS: s_cost[i][j]
    const int j = e / n, i = e % n;
    float v = c[static_cast<size_t>(j) * ngt + i];
    // A diverged model hands in NaN / inf costs.  scipy.optimize.linear_sum_assignment raises ValueError on them
    // ("matrix contains invalid numeric entries"), which stops the reference's training; a kernel cannot raise, so the
    // search runs on clamped costs (it always finds a column and terminates) and the problem is POISONED below: its
    // matched mask becomes NaN, the loss of the step becomes NaN, and engine.py:155-157 stops the run.
    bad |= !(fabsf(v) <= FLT_MAX);
    if (!(v == v)) v = FLT_MAX;
    v = fminf(fmaxf(v, -FLT_MAX), FLT_MAX);
    s_cost[i * m + j] = v;
  }
  if (bad) s_invalid = 1;
  if (tid < n) {
    s_u[tid] = 0.0;
    s_col4row[tid] = -1;
  }
  __syncthreads();

  for (int cur = 0; cur < n; ++cur) {
    for (int j = tid; j < m; j += kHungThreads) {
      s_spc[j] = DBL_MAX;
      s_path[j] = -1;
      s_insc[j] = 0;
    }
    if (tid < n) s_insr[tid] = 0;
    __syncthreads();
    int i = cur, sink = -1;
    double min_val = 0.0;
    for (int guard = 0; sink < 0 && guard <= m; ++guard) {  // block-uniform: every thread follows the same (i, min_val, sink); <= m steps
      if (tid == 0) s_insr[i] = 1;
      const double ui = s_u[i];
      Best mine{DBL_MAX, 2, 0x7fffffff};
      for (int j = tid; j < m; j += kHungThreads) {
        if (s_insc[j]) continue;
        const double r = min_val + static_cast<double>(s_cost[i * m + j]) - ui - s_v[j];
        if (r < s_spc[j]) {
          s_spc[j] = r;
          s_path[j] = i;
        }
        const Best cand{s_spc[j], s_row4col[j] >= 0 ? 1 : 0, j};
        if (better(cand, mine)) mine = cand;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.val = __shfl_xor(mine.val, off);
        o.taken = __shfl_xor(mine.taken, off);
        o.col = __shfl_xor(mine.col, off);
        if (better(o, mine)) mine = o;
      }
      if (lane == 0) s_best[w] = mine;
      __syncthreads();
      if (tid == 0) {
        Best b = s_best[0];
        for (int q = 1; q < kHungThreads / kWave; ++q)
          if (better(s_best[q], b)) b = s_best[q];
        s_pick = b;
        if (b.col < m) s_insc[b.col] = 1;
      }
      __syncthreads();
      const Best pick = s_pick;
      if (pick.col >= m) break;  // cannot happen with finite costs and n <= m
      min_val = pick.val;
      const int owner = s_row4col[pick.col];
      if (owner < 0) sink = pick.col;
      else i = owner;
    }
    if (sink < 0) continue;  // (see above) leave this row unassigned
    // dual update (rows and columns that were reached), then flip the augmenting path
    if (tid < n && s_insr[tid]) {
      if (tid == cur) s_u[tid] += min_val;
      else s_u[tid] += min_val - s_spc[s_col4row[tid]];
    }
    for (int j = tid; j < m; j += kHungThreads)
      if (s_insc[j]) s_v[j] -= min_val - s_spc[j];
    __syncthreads();
    if (tid == 0) {
      int j = sink;
      for (;;) {
        const int r = s_path[j];
        s_row4col[j] = r;
        const int prev = s_col4row[r];
        s_col4row[r] = j;
        j = prev;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  if (tid < n) {
    const int j = s_col4row[tid];
    if (j >= 0) {
      inds[j] = tid;
      mask[j] = 1.0f;
    }
  }
  if (s_invalid) {  // block-uniform (written before the first barrier of the search)
    __syncthreads();
    for (int j = tid; j < m; j += kHungThreads) mask[j] = __builtin_nanf("");
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_hungarian_f32(const float *cost, const int64_t *nactual, int64_t *per_prop_gt_inds, float *matched_mask,
                                int nprob, int nq, int ngt, void *stream) {
  using namespace coda;
  if (nprob < 0 || nq < 0 || ngt < 0) return CODA_EINVAL;
  if (nprob == 0 || nq == 0) return CODA_OK;
  if (!cost || !nactual || !per_prop_gt_inds || !matched_mask) return CODA_EINVAL;
  const size_t lds = static_cast<size_t>(nq) * (8 + 8 + 4 + 4 + 4) + sizeof(float) * static_cast<size_t>(nq) * ngt;
  if (nq > 1024 || ngt > kMaxRows || nq < ngt || lds > 150 * 1024) return CODA_ENOSPC;
  auto kern = hungarian_kernel;
  if (int st = raise_dynamic_lds(kern, lds); st != CODA_OK) return st;  // CODA_ENOSPC: solver="auto" takes the host route
  clear_sticky_error();
  hipLaunchKernelGGL(kern, dim3(nprob), dim3(kHungThreads), lds, static_cast<hipStream_t>(stream), cost, nactual,
                     per_prop_gt_inds, matched_mask, nq, ngt);
  return launch_status_nospace();
}
