// sa_mlp.hip -- streaming kernels of the set-abstraction shared MLP (gfx950).
//
// The reference runs SharedMLP([3,64,128,256]) as 1x1 Conv2d + BatchNorm2d + in-place
// ReLU per layer and F.max_pool2d over the nsample axis (pytorch_utils.py:8-33,
// pointnet2_modules.py:247-253) on (B,C,M,S) tensors: per layer a conv output, a BN
// read+write, a ReLU pass, layout transposes inside MIOpen, and a (1 x S) max-pool that
// runs at 0.3 TB/s.  Here the activations are kept CHANNELS-LAST, (P = B*M*S rows, C),
// the 1x1 convolutions are plain library GEMMs on that matrix, and everything around
// them is fused into a few HBM-streaming kernels:
//   fwd: l1_stats / l1_apply   (3->C1 conv recomputed on the fly, never stored pre-BN)
//        col_stats             (per-channel sum, sum of squares -> batch statistics)
//        bn_relu_apply         (scale/shift + ReLU)
//        col_stats_pool        (last layer: statistics AND per-(centre,channel) max/min/arg
//                               over the S samples in one read; BN+ReLU are monotone per
//                               channel, so pooling the pre-BN values is exact)
//   bwd: bn_bwd_sparse         (max-pool + ReLU + BN backward of the last layer in one pass)
//        relu_bn_bwd_stats / relu_bn_bwd_apply, l1_bwd_stats / l1_bwd_dw
// Batch statistics are accumulated in fp64 (block partials in fp32), one atomic per block
// and channel.  All kernels: 256 threads, a thread owns 4 consecutive channels (float4),
// C/4 threads cover a row, so every load/store is a full-line coalesced access.
#include "coda_sa_mlp.h"
#include "common.hip.h"

namespace coda {
namespace {

constexpr int kT = 256;
constexpr int kMaxBlocks = 512;  // streaming kernels: ~2 workgroups per CU, each ends in 2*C fp64 atomics

struct RowMap {
  int tpr, rpb, cq, rsub;  // threads per row, rows per pass, channel quad, row sub-index
};
__device__ __forceinline__ RowMap row_map(int c) {
  RowMap m;
  m.tpr = c >> 2;
  m.rpb = kT / m.tpr;
  m.cq = threadIdx.x % m.tpr;
  m.rsub = threadIdx.x / m.tpr;
  return m;
}

// rows streamed by one block: the grid (<= kMaxBlocks) splits the p rows evenly, in whole passes
__device__ __forceinline__ long long block_rows(long long p, int rpb) {
  const long long per = (p + gridDim.x - 1) / gridDim.x;
  return (per + rpb - 1) / rpb * rpb;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// Sum `nacc` float4 accumulators over the threads that share a channel quad and add the
// result to sums[k*C + 4*cq .. +3] (double) with one atomic per channel and block.
template <int NACC>
__device__ __forceinline__ void block_reduce_to_global(const float4 (&acc)[NACC], const RowMap &m, int c,
                                                       double *sums) {
  __shared__ float4 s_part[NACC][kT];
#pragma unroll
  for (int k = 0; k < NACC; ++k) s_part[k][threadIdx.x] = acc[k];
  __syncthreads();
  if (m.rsub == 0) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      for (int r = 0; r < m.rpb; ++r) {
        const float4 v = s_part[k][r * m.tpr + m.cq];
        t0 += v.x; t1 += v.y; t2 += v.z; t3 += v.w;
      }
      double *dst = sums + static_cast<size_t>(k) * c + 4 * m.cq;
      atomicAdd(dst + 0, t0); atomicAdd(dst + 1, t1); atomicAdd(dst + 2, t2); atomicAdd(dst + 3, t3);
    }
  }
}

// y[c] = x . w1[c]  for the thread's 4 channels (w1 is (C,3) row-major)
struct W4 { float4 w0, w1, w2; };  // component k of the 4 channels
__device__ __forceinline__ W4 load_w1(const float *w1, int cq) {
  const float *p = w1 + 12 * cq;
  W4 w;
  w.w0 = make_float4(p[0], p[3], p[6], p[9]);
  w.w1 = make_float4(p[1], p[4], p[7], p[10]);
  w.w2 = make_float4(p[2], p[5], p[8], p[11]);
  return w;
}
__device__ __forceinline__ float4 conv3(const float *x, const W4 &w) {
  const float x0 = x[0], x1 = x[1], x2 = x[2];
  return make_float4(x0 * w.w0.x + x1 * w.w1.x + x2 * w.w2.x, x0 * w.w0.y + x1 * w.w1.y + x2 * w.w2.y,
                     x0 * w.w0.z + x1 * w.w1.z + x2 * w.w2.z, x0 * w.w0.w + x1 * w.w1.w + x2 * w.w2.w);
}
__device__ __forceinline__ float4 affine(float4 y, float4 a, float4 b) {
  return make_float4(y.x * a.x + b.x, y.y * a.y + b.y, y.z * a.z + b.z, y.w * a.w + b.w);
}
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void acc_stats(float4 (&acc)[2], float4 y) {
  acc[0].x += y.x; acc[0].y += y.y; acc[0].z += y.z; acc[0].w += y.w;
  acc[1].x += y.x * y.x; acc[1].y += y.y * y.y; acc[1].z += y.z * y.z; acc[1].w += y.w * y.w;
}
// row counted `w` times (de-duplicated groups: the first row of a group stands for its copies)
__device__ __forceinline__ void acc_stats_w(float4 (&acc)[2], float4 y, float w) {
  const float4 wy = make_float4(w * y.x, w * y.y, w * y.z, w * y.w);
  acc[0].x += wy.x; acc[0].y += wy.y; acc[0].z += wy.z; acc[0].w += wy.w;
  acc[1].x += wy.x * y.x; acc[1].y += wy.y * y.y; acc[1].z += wy.z * y.z; acc[1].w += wy.w * y.w;
}

// ---- forward ---------------------------------------------------------------------------
template <bool FROM_X>
__global__ __launch_bounds__(kT) void col_stats_kernel(const float *__restrict__ src,
                                                       const float *__restrict__ w1, long long p, int c,
                                                       const float *__restrict__ roww,
                                                       double *__restrict__ sums) {
  const RowMap m = row_map(c);
  W4 w;
  if (FROM_X) w = load_w1(w1, m.cq);
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  const long long rows_pb = block_rows(p, m.rpb);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_pb;
  const long long r1 = min(r0 + rows_pb, p);
  for (long long r = r0 + m.rsub; r < r1; r += m.rpb) {
    const float4 y = FROM_X ? conv3(src + r * 3, w) : ld4(src + r * c + 4 * m.cq);
    if (roww) acc_stats_w(acc, y, roww[r]);
    else acc_stats(acc, y);
  }
  block_reduce_to_global<2>(acc, m, c, sums);
}

template <bool FROM_X>
__global__ __launch_bounds__(kT) void bn_relu_apply_kernel(const float *__restrict__ src,
                                                           const float *__restrict__ w1,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, long long p,
                                                           int c, float *__restrict__ dst) {
  const RowMap m = row_map(c);
  W4 w;
  if (FROM_X) w = load_w1(w1, m.cq);
  const float4 a = ld4(scale + 4 * m.cq), b = ld4(shift + 4 * m.cq);
  const long long rows_pb = block_rows(p, m.rpb);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_pb;
  const long long r1 = min(r0 + rows_pb, p);
  for (long long r = r0 + m.rsub; r < r1; r += m.rpb) {
    const float4 y = FROM_X ? conv3(src + r * 3, w) : ld4(src + r * c + 4 * m.cq);
    st4(dst + r * c + 4 * m.cq, relu4(affine(y, a, b)));
  }
}

// Last layer: statistics + max / min / arg over the S rows of each group (centre).
__global__ __launch_bounds__(kT) void col_stats_pool_kernel(const float *__restrict__ y, long long groups,
                                                            int s_fixed, int c, const float *__restrict__ roww,
                                                            const int *__restrict__ goff,
                                                            double *__restrict__ sums,
                                                            float *__restrict__ ymax, float *__restrict__ ymin,
                                                            int *__restrict__ amax, int *__restrict__ amin) {
  __shared__ float4 s_mx[kT], s_mn[kT];
  __shared__ int4 s_ax[kT], s_an[kT];
  const RowMap m = row_map(c);
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long long row0 = goff ? goff[g] : g * s_fixed;
    const int s = goff ? goff[g + 1] - goff[g] : s_fixed;
    const float *base = y + row0 * c + 4 * m.cq;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    int4 ax = make_int4(0, 0, 0, 0), an = make_int4(0, 0, 0, 0);
    for (int r = m.rsub; r < s; r += m.rpb) {
      const float4 v = ld4(base + static_cast<long long>(r) * c);
      if (roww) acc_stats_w(acc, v, roww[row0 + r]);
      else acc_stats(acc, v);
      if (v.x > mx.x) { mx.x = v.x; ax.x = r; }
      if (v.y > mx.y) { mx.y = v.y; ax.y = r; }
      if (v.z > mx.z) { mx.z = v.z; ax.z = r; }
      if (v.w > mx.w) { mx.w = v.w; ax.w = r; }
      if (v.x < mn.x) { mn.x = v.x; an.x = r; }
      if (v.y < mn.y) { mn.y = v.y; an.y = r; }
      if (v.z < mn.z) { mn.z = v.z; an.z = r; }
      if (v.w < mn.w) { mn.w = v.w; an.w = r; }
    }
    s_mx[threadIdx.x] = mx; s_mn[threadIdx.x] = mn; s_ax[threadIdx.x] = ax; s_an[threadIdx.x] = an;
    __syncthreads();
    if (m.rsub == 0) {
      for (int q = 1; q < m.rpb; ++q) {  // ascending rsub; strict compare keeps the lowest sample on ties
        const float4 ox = s_mx[q * m.tpr + m.cq], on = s_mn[q * m.tpr + m.cq];
        const int4 oax = s_ax[q * m.tpr + m.cq], oan = s_an[q * m.tpr + m.cq];
        if (ox.x > mx.x || (ox.x == mx.x && oax.x < ax.x)) { mx.x = ox.x; ax.x = oax.x; }
        if (ox.y > mx.y || (ox.y == mx.y && oax.y < ax.y)) { mx.y = ox.y; ax.y = oax.y; }
        if (ox.z > mx.z || (ox.z == mx.z && oax.z < ax.z)) { mx.z = ox.z; ax.z = oax.z; }
        if (ox.w > mx.w || (ox.w == mx.w && oax.w < ax.w)) { mx.w = ox.w; ax.w = oax.w; }
        if (on.x < mn.x || (on.x == mn.x && oan.x < an.x)) { mn.x = on.x; an.x = oan.x; }
        if (on.y < mn.y || (on.y == mn.y && oan.y < an.y)) { mn.y = on.y; an.y = oan.y; }
        if (on.z < mn.z || (on.z == mn.z && oan.z < an.z)) { mn.z = on.z; an.z = oan.z; }
        if (on.w < mn.w || (on.w == mn.w && oan.w < an.w)) { mn.w = on.w; an.w = oan.w; }
      }
      const long long o = g * c + 4 * m.cq;
      st4(ymax + o, mx);
      st4(ymin + o, mn);
      *reinterpret_cast<int4 *>(amax + o) = ax;
      *reinterpret_cast<int4 *>(amin + o) = an;
    }
    __syncthreads();
  }
  block_reduce_to_global<2>(acc, m, c, sums);
}

// C == 256: a row is exactly one wave wide, so each of the 4 waves of a block walks its own group
// (4 row loads in flight) with no LDS exchange or barrier per group -- with de-duplicated groups
// of ~16 rows the per-group synchronisation of the kernel above costs more than the data.
__device__ __forceinline__ void pool_step(float4 v, int r, float4 &mx, float4 &mn, int4 &ax, int4 &an) {
  if (v.x > mx.x) { mx.x = v.x; ax.x = r; }
  if (v.y > mx.y) { mx.y = v.y; ax.y = r; }
  if (v.z > mx.z) { mx.z = v.z; ax.z = r; }
  if (v.w > mx.w) { mx.w = v.w; ax.w = r; }
  if (v.x < mn.x) { mn.x = v.x; an.x = r; }
  if (v.y < mn.y) { mn.y = v.y; an.y = r; }
  if (v.z < mn.z) { mn.z = v.z; an.z = r; }
  if (v.w < mn.w) { mn.w = v.w; an.w = r; }
}
__global__ __launch_bounds__(kT) void col_stats_pool_wave_kernel(const float *__restrict__ y, long long groups,
                                                                 int s_fixed, const float *__restrict__ roww,
                                                                 const int *__restrict__ goff,
                                                                 double *__restrict__ sums,
                                                                 float *__restrict__ ymax, float *__restrict__ ymin,
                                                                 int *__restrict__ amax, int *__restrict__ amin) {
  constexpr int c = 256;
  const RowMap m = row_map(c);  // tpr = 64: cq = lane, rsub = wave
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  for (long long g = static_cast<long long>(blockIdx.x) * 4 + m.rsub; g < groups;
       g += static_cast<long long>(gridDim.x) * 4) {
    const long long row0 = goff ? goff[g] : g * s_fixed;
    const int s = goff ? goff[g + 1] - goff[g] : s_fixed;
    const float *base = y + row0 * c + 4 * m.cq;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    int4 ax = make_int4(0, 0, 0, 0), an = make_int4(0, 0, 0, 0);
    int r = 0;
    for (; r + 4 <= s; r += 4) {
      const float4 v0 = ld4(base + static_cast<long long>(r) * c), v1 = ld4(base + static_cast<long long>(r + 1) * c),
                   v2 = ld4(base + static_cast<long long>(r + 2) * c), v3 = ld4(base + static_cast<long long>(r + 3) * c);
      if (roww) {
        acc_stats_w(acc, v0, roww[row0 + r]); acc_stats_w(acc, v1, roww[row0 + r + 1]);
        acc_stats_w(acc, v2, roww[row0 + r + 2]); acc_stats_w(acc, v3, roww[row0 + r + 3]);
      } else {
        acc_stats(acc, v0); acc_stats(acc, v1); acc_stats(acc, v2); acc_stats(acc, v3);
      }
      pool_step(v0, r, mx, mn, ax, an); pool_step(v1, r + 1, mx, mn, ax, an);
      pool_step(v2, r + 2, mx, mn, ax, an); pool_step(v3, r + 3, mx, mn, ax, an);
    }
    for (; r < s; ++r) {
      const float4 v = ld4(base + static_cast<long long>(r) * c);
      if (roww) acc_stats_w(acc, v, roww[row0 + r]);
      else acc_stats(acc, v);
      pool_step(v, r, mx, mn, ax, an);
    }
    const long long o = g * c + 4 * m.cq;
    st4(ymax + o, mx);
    st4(ymin + o, mn);
    *reinterpret_cast<int4 *>(amax + o) = ax;
    *reinterpret_cast<int4 *>(amin + o) = an;
  }
  block_reduce_to_global<2>(acc, m, c, sums);
}

// ---- backward --------------------------------------------------------------------------
// Last layer.  d (G,C) is the gradient that survived max-pool + ReLU, living at sample
// sel[g][c] of its group; BN backward makes it dense:
//   dy[p][c] = coef_a[c] * ( (row-in-group == sel ? d : 0) - m1[c] - xhat[p][c] * m2[c] )
__global__ __launch_bounds__(kT) void bn_bwd_sparse_kernel(const float *__restrict__ y,
                                                           const float *__restrict__ d,
                                                           const int *__restrict__ sel,
                                                           const float *__restrict__ coef,  // [5][C]
                                                           long long groups, int s_fixed, int c,
                                                           const float *__restrict__ roww,
                                                           const int *__restrict__ goff,
                                                           float *__restrict__ dy) {
  const RowMap m = row_map(c);
  const float4 a = ld4(coef + 4 * m.cq), m1 = ld4(coef + c + 4 * m.cq), m2 = ld4(coef + 2 * c + 4 * m.cq),
               mu = ld4(coef + 3 * c + 4 * m.cq), is = ld4(coef + 4 * c + 4 * m.cq);
  for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long long o = g * c + 4 * m.cq;
    const float4 dg = ld4(d + o);
    const int4 sg = *reinterpret_cast<const int4 *>(sel + o);
    const long long row0 = goff ? goff[g] : g * s_fixed;
    const int s = goff ? goff[g + 1] - goff[g] : s_fixed;
    for (int r = m.rsub; r < s; r += m.rpb) {
      const long long off = (row0 + r) * c + 4 * m.cq;
      const float4 v = ld4(y + off);
      const float w = roww ? roww[row0 + r] : 1.0f;  // the row's gradient stands for w identical rows
      float4 out;
      out.x = a.x * ((r == sg.x ? dg.x : 0.f) - w * (m1.x + (v.x - mu.x) * is.x * m2.x));
      out.y = a.y * ((r == sg.y ? dg.y : 0.f) - w * (m1.y + (v.y - mu.y) * is.y * m2.y));
      out.z = a.z * ((r == sg.z ? dg.z : 0.f) - w * (m1.z + (v.z - mu.z) * is.z * m2.z));
      out.w = a.w * ((r == sg.w ? dg.w : 0.f) - w * (m1.w + (v.w - mu.w) * is.w * m2.w));
      st4(dy + off, out);
    }
  }
}

// Hidden layers: d = da where BN(y) > 0 else 0.  params: [4][C] = scale, shift, mean, invstd.
template <bool FROM_X>
__global__ __launch_bounds__(kT) void relu_bn_bwd_stats_kernel(const float *__restrict__ da,
                                                               const float *__restrict__ src,
                                                               const float *__restrict__ w1,
                                                               const float *__restrict__ prm, long long p,
                                                               int c, double *__restrict__ sums) {
  const RowMap m = row_map(c);
  W4 w;
  if (FROM_X) w = load_w1(w1, m.cq);
  const float4 a = ld4(prm + 4 * m.cq), b = ld4(prm + c + 4 * m.cq), mu = ld4(prm + 2 * c + 4 * m.cq),
               is = ld4(prm + 3 * c + 4 * m.cq);
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  const long long rows_pb = block_rows(p, m.rpb);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_pb;
  const long long r1 = min(r0 + rows_pb, p);
  for (long long r = r0 + m.rsub; r < r1; r += m.rpb) {
    const float4 y = FROM_X ? conv3(src + r * 3, w) : ld4(src + r * c + 4 * m.cq);
    const float4 g = ld4(da + r * c + 4 * m.cq);
    const float4 act = affine(y, a, b);
    const float dx = act.x > 0.f ? g.x : 0.f, dyv = act.y > 0.f ? g.y : 0.f, dz = act.z > 0.f ? g.z : 0.f,
                dw = act.w > 0.f ? g.w : 0.f;
    acc[0].x += dx; acc[0].y += dyv; acc[0].z += dz; acc[0].w += dw;
    acc[1].x += dx * (y.x - mu.x) * is.x; acc[1].y += dyv * (y.y - mu.y) * is.y;
    acc[1].z += dz * (y.z - mu.z) * is.z; acc[1].w += dw * (y.w - mu.w) * is.w;
  }
  block_reduce_to_global<2>(acc, m, c, sums);
}

// dy = coef_a * (d - m1 - xhat * m2).  prm: [7][C] = scale, shift, mean, invstd, coef_a, m1, m2.
// DW (layer 1 only): instead of storing dy, accumulate dW1[c][k] = sum_p dy[p][c] * x[p][k].
template <bool FROM_X>
__global__ __launch_bounds__(kT) void relu_bn_bwd_apply_kernel(const float *__restrict__ da,
                                                               const float *__restrict__ src,
                                                               const float *__restrict__ w1,
                                                               const float *__restrict__ prm, long long p,
                                                               int c, const float *__restrict__ roww,
                                                               float *__restrict__ dy,
                                                               double *__restrict__ dw1) {
  const RowMap m = row_map(c);
  W4 w;
  if (FROM_X) w = load_w1(w1, m.cq);
  const float4 a = ld4(prm + 4 * m.cq), b = ld4(prm + c + 4 * m.cq), mu = ld4(prm + 2 * c + 4 * m.cq),
               is = ld4(prm + 3 * c + 4 * m.cq), ca = ld4(prm + 4 * c + 4 * m.cq),
               m1 = ld4(prm + 5 * c + 4 * m.cq), m2 = ld4(prm + 6 * c + 4 * m.cq);
  float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  const long long rows_pb = block_rows(p, m.rpb);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_pb;
  const long long r1 = min(r0 + rows_pb, p);
  for (long long r = r0 + m.rsub; r < r1; r += m.rpb) {
    const float4 y = FROM_X ? conv3(src + r * 3, w) : ld4(src + r * c + 4 * m.cq);
    const float4 g = ld4(da + r * c + 4 * m.cq);
    const float4 act = affine(y, a, b);
    const float wr = roww ? roww[r] : 1.0f;
    float4 out;
    out.x = ca.x * ((act.x > 0.f ? g.x : 0.f) - wr * (m1.x + (y.x - mu.x) * is.x * m2.x));
    out.y = ca.y * ((act.y > 0.f ? g.y : 0.f) - wr * (m1.y + (y.y - mu.y) * is.y * m2.y));
    out.z = ca.z * ((act.z > 0.f ? g.z : 0.f) - wr * (m1.z + (y.z - mu.z) * is.z * m2.z));
    out.w = ca.w * ((act.w > 0.f ? g.w : 0.f) - wr * (m1.w + (y.w - mu.w) * is.w * m2.w));
    if (FROM_X) {
      const float x0 = src[r * 3], x1 = src[r * 3 + 1], x2 = src[r * 3 + 2];
      acc[0].x += out.x * x0; acc[0].y += out.y * x0; acc[0].z += out.z * x0; acc[0].w += out.w * x0;
      acc[1].x += out.x * x1; acc[1].y += out.y * x1; acc[1].z += out.z * x1; acc[1].w += out.w * x1;
      acc[2].x += out.x * x2; acc[2].y += out.y * x2; acc[2].z += out.z * x2; acc[2].w += out.w * x2;
    } else {
      st4(dy + r * c + 4 * m.cq, out);
    }
  }
  if (FROM_X) block_reduce_to_global<3>(acc, m, c, dw1);  // dw1 laid out [3][C]
}

bool bad_c(int c) { return c < 4 || c > 1024 || (c % 4) != 0 || (kT % (c / 4)) != 0; }
int nblocks(long long p) {
  const long long want = (p + 255) / 256;  // at least 256 rows per block
  return static_cast<int>(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
}


// ---- batch-norm bookkeeping (what used to be ~20 tiny framework kernels per layer and pass) --------------
// sums = [sum y, sum y^2] over n rows (double).  stats = [scale, shift, mean, invstd][C] (float):
//   mean = s1 / n, var = max(s2 / n - mean^2, 0) in double; invstd = 1 / sqrt(var + eps);
//   scale = gamma * invstd, shift = beta - mean * scale (float, one rounding per operation);
//   running_mean = running_mean * (1 - mom) + mom * mean, running_var likewise with the unbiased variance
//   var * n / max(n - 1, 1) (torch.nn.BatchNorm's update), num_batches += 1.
__global__ void bn_finalize_kernel(const double *__restrict__ sums, double n, double eps, float momentum,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   long long *__restrict__ num_batches, float *__restrict__ stats, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && num_batches) *num_batches += 1;
  if (i >= c) return;
  const double mean = sums[i] / n;
  double var = sums[c + i] / n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float mean32 = static_cast<float>(mean);
  const float invstd = static_cast<float>(1.0 / sqrt(var + eps));
  const float scale = __fmul_rn(gamma[i], invstd);
  stats[i] = scale;
  stats[c + i] = __fsub_rn(beta[i], __fmul_rn(mean32, scale));
  stats[2 * c + i] = mean32;
  stats[3 * c + i] = invstd;
  if (running_mean) {
    const float keep = 1.0f - momentum;
    const double nm1 = n - 1.0 > 1.0 ? n - 1.0 : 1.0;
    const float unbiased = static_cast<float>(var * (n / nm1));
    running_mean[i] = __fadd_rn(__fmul_rn(running_mean[i], keep), __fmul_rn(momentum, mean32));
    running_var[i] = __fadd_rn(__fmul_rn(running_var[i], keep), __fmul_rn(momentum, unbiased));
  }
}

// sums = [sum d, sum d * xhat] (double).  dbeta / dgamma = float(sums) (may be NULL);
// coef (may be NULL): layout 0 = [scale, shift, mean, invstd, gamma * invstd, m1, m2][C] (hidden layers),
// layout 1 = [gamma * invstd, m1, m2, mean, invstd][C] (pooled last layer), m1 = sum d / n, m2 = sum d xhat / n
// (n <= 0: eval mode, m1 = m2 = 0).
__global__ void bn_bwd_coef_kernel(const double *__restrict__ sums, double n, const float *__restrict__ gamma,
                                   const float *__restrict__ stats, float *__restrict__ coef, int layout,
                                   float *__restrict__ dbeta, float *__restrict__ dgamma, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  if (dbeta) dbeta[i] = static_cast<float>(sums[i]);
  if (dgamma) dgamma[i] = static_cast<float>(sums[c + i]);
  if (!coef) return;
  const float m1 = n > 0.0 ? static_cast<float>(sums[i] / n) : 0.0f;
  const float m2 = n > 0.0 ? static_cast<float>(sums[c + i] / n) : 0.0f;
  const float mean = stats[2 * c + i], invstd = stats[3 * c + i];
  const float a = __fmul_rn(gamma[i], invstd);
  if (layout == 0) {
    coef[i] = stats[i]; coef[c + i] = stats[c + i]; coef[2 * c + i] = mean; coef[3 * c + i] = invstd;
    coef[4 * c + i] = a; coef[5 * c + i] = m1; coef[6 * c + i] = m2;
  } else {
    coef[i] = a; coef[c + i] = m1; coef[2 * c + i] = m2; coef[3 * c + i] = mean; coef[4 * c + i] = invstd;
  }
}

// Pooled last layer, forward: BN is monotone per channel, so the max-pool of relu(bn(y)) is relu(bn(.)) of the
// group's max (scale >= 0) or min (scale < 0) pre-BN value.
__global__ __launch_bounds__(kT) void pool_select_kernel(const float *__restrict__ ymax, const float *__restrict__ ymin,
                                                         const int32_t *__restrict__ amax, const int32_t *__restrict__ amin,
                                                         const float *__restrict__ stats, float *__restrict__ ysel,
                                                         int32_t *__restrict__ sel, float *__restrict__ out,
                                                         long long total, int c) {
  for (long long i = blockIdx.x * static_cast<long long>(kT) + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * kT) {
    const int ch = static_cast<int>(i % c);
    const float scale = stats[ch], shift = stats[c + ch];
    const bool pos = scale >= 0.0f;
    const float y = pos ? ymax[i] : ymin[i];
    ysel[i] = y;
    sel[i] = pos ? amax[i] : amin[i];
    out[i] = fmaxf(__fadd_rn(__fmul_rn(y, scale), shift), 0.0f);
  }
}

// Pooled last layer, backward: d = gout where out > 0; sums = [sum d, sum d * (ysel - mean) * invstd] (double).
__global__ __launch_bounds__(kT) void pool_bwd_stats_kernel(const float *__restrict__ gout, const float *__restrict__ out,
                                                            const float *__restrict__ ysel, const float *__restrict__ stats,
                                                            float *__restrict__ d, long long groups, int c,
                                                            double *__restrict__ sums) {
  const RowMap m = row_map(c);
  const float4 mu = ld4(stats + 2 * c + 4 * m.cq), is = ld4(stats + 3 * c + 4 * m.cq);
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  const long long rows_pb = block_rows(groups, m.rpb);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_pb;
  const long long r1 = min(r0 + rows_pb, groups);
  auto one = [&](long long at, float4 g, float4 o, float4 y) {
    const float4 dv = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
    st4(d + at, dv);
    acc[0].x += dv.x; acc[0].y += dv.y; acc[0].z += dv.z; acc[0].w += dv.w;
    acc[1].x += dv.x * ((y.x - mu.x) * is.x); acc[1].y += dv.y * ((y.y - mu.y) * is.y);
    acc[1].z += dv.z * ((y.z - mu.z) * is.z); acc[1].w += dv.w * ((y.w - mu.w) * is.w);
  };
  constexpr int U = 4;  // rows in flight per thread (three input streams each); same accumulation order
  long long r = r0 + m.rsub;
  for (; r + (U - 1) * m.rpb < r1; r += U * m.rpb) {
    float4 g[U], o[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long at = (r + u * m.rpb) * c + 4 * m.cq;
      g[u] = ld4(gout + at); o[u] = ld4(out + at); y[u] = ld4(ysel + at);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one((r + u * m.rpb) * c + 4 * m.cq, g[u], o[u], y[u]);
  }
  for (; r < r1; r += m.rpb) {
    const long long at = r * c + 4 * m.cq;
    one(at, ld4(gout + at), ld4(out + at), ld4(ysel + at));
  }
  block_reduce_to_global<2>(acc, m, c, sums);
}

// one thread per (group, slot) and, behind them, per padding row
__global__ __launch_bounds__(256) void compact_groups_kernel(const float *__restrict__ grouped, const int64_t *__restrict__ cnt,
                                                             const int64_t *__restrict__ goff, float *__restrict__ x,
                                                             float *__restrict__ roww, int32_t *__restrict__ goff32,
                                                             long long groups, int s_len, long long total,
                                                             long long rows_padded) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long slots = groups * s_len;
  if (i < slots) {
    const long long g = i / s_len;
    const int j = static_cast<int>(i - g * s_len);
    const long long c = cnt[g], base = goff[g];
    if (j < c) {
      const float *src = grouped + i * 3;
      float *dst = x + (base + j) * 3;
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
      roww[base + j] = j == 0 ? static_cast<float>(s_len - c + 1) : 1.0f;
    }
    if (j == 0) goff32[g] = static_cast<int32_t>(base);
    if (i == 0) goff32[groups] = static_cast<int32_t>(total);
  } else {
    const long long r = total + (i - slots);
    if (r < rows_padded) {
      x[r * 3] = 0.f; x[r * 3 + 1] = 0.f; x[r * 3 + 2] = 0.f;
      roww[r] = 0.f;
    }
  }
}

}  // namespace
}  // namespace coda

using namespace coda;

CODA_API int coda_sa_col_stats_f32(const float *src, const float *w1, long long p, int c, const float *row_weight,
                                   double *sums, void *stream) {
  if (p < 0 || bad_c(c) || !sums) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (p == 0) return CODA_OK;
  if (!src) return CODA_EINVAL;
  clear_sticky_error();
  if (w1) hipLaunchKernelGGL(col_stats_kernel<true>, dim3(nblocks(p)), dim3(kT), 0, s, src, w1, p, c, row_weight, sums);
  else hipLaunchKernelGGL(col_stats_kernel<false>, dim3(nblocks(p)), dim3(kT), 0, s, src, w1, p, c, row_weight, sums);
  return launch_status();
}

CODA_API int coda_sa_bn_relu_apply_f32(const float *src, const float *w1, const float *scale,
                                       const float *shift, long long p, int c, float *dst, void *stream) {
  if (p < 0 || bad_c(c)) return CODA_EINVAL;
  if (p == 0) return CODA_OK;
  if (!src || !scale || !shift || !dst) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  if (w1) hipLaunchKernelGGL(bn_relu_apply_kernel<true>, dim3(nblocks(p)), dim3(kT), 0, s, src, w1, scale, shift, p, c, dst);
  else hipLaunchKernelGGL(bn_relu_apply_kernel<false>, dim3(nblocks(p)), dim3(kT), 0, s, src, w1, scale, shift, p, c, dst);
  return launch_status();
}

CODA_API int coda_sa_col_stats_pool_f32(const float *y, long long groups, int s_len, int c,
                                        const float *row_weight, const int32_t *group_offsets, double *sums,
                                        float *ymax, float *ymin, int32_t *amax, int32_t *amin,
                                        void *stream) {
  if (groups < 0 || (s_len <= 0 && !group_offsets) || bad_c(c) || !sums) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (groups == 0) return CODA_OK;
  if (!y || !ymax || !ymin || !amax || !amin) return CODA_EINVAL;
  const int grid = static_cast<int>(groups < 4096 ? groups : 4096);
  clear_sticky_error();
  if (c == 256) {
    const long long want = (groups + 3) / 4;
    hipLaunchKernelGGL(col_stats_pool_wave_kernel, dim3(static_cast<int>(want < 2048 ? want : 2048)), dim3(kT), 0, s,
                       y, groups, s_len, row_weight, group_offsets, sums, ymax, ymin, amax, amin);
  } else {
    hipLaunchKernelGGL(col_stats_pool_kernel, dim3(grid), dim3(kT), 0, s, y, groups, s_len, c, row_weight,
                       group_offsets, sums, ymax, ymin, amax, amin);
  }
  return launch_status();
}

CODA_API int coda_sa_bn_bwd_sparse_f32(const float *y, const float *d, const int32_t *sel, const float *coef,
                                       long long groups, int s_len, int c, const float *row_weight,
                                       const int32_t *group_offsets, float *dy, void *stream) {
  if (groups < 0 || (s_len <= 0 && !group_offsets) || bad_c(c)) return CODA_EINVAL;
  if (groups == 0) return CODA_OK;
  if (!y || !d || !sel || !coef || !dy) return CODA_EINVAL;
  const int grid = static_cast<int>(groups < 8192 ? groups : 8192);
  clear_sticky_error();
  hipLaunchKernelGGL(bn_bwd_sparse_kernel, dim3(grid), dim3(kT), 0, static_cast<hipStream_t>(stream), y, d, sel,
                     coef, groups, s_len, c, row_weight, group_offsets, dy);
  return launch_status();
}

CODA_API int coda_sa_relu_bn_bwd_stats_f32(const float *da, const float *src, const float *w1,
                                           const float *prm, long long p, int c, double *sums,
                                           void *stream) {
  if (p < 0 || bad_c(c) || !sums) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (p == 0) return CODA_OK;
  if (!da || !src || !prm) return CODA_EINVAL;
  clear_sticky_error();
  if (w1) hipLaunchKernelGGL(relu_bn_bwd_stats_kernel<true>, dim3(nblocks(p)), dim3(kT), 0, s, da, src, w1, prm, p, c, sums);
  else hipLaunchKernelGGL(relu_bn_bwd_stats_kernel<false>, dim3(nblocks(p)), dim3(kT), 0, s, da, src, w1, prm, p, c, sums);
  return launch_status();
}

CODA_API int coda_sa_relu_bn_bwd_apply_f32(const float *da, const float *src, const float *w1,
                                           const float *prm, long long p, int c, const float *row_weight,
                                           float *dy, double *dw1, void *stream) {
  if (p < 0 || bad_c(c)) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (w1) {
    if (!dw1) return CODA_EINVAL;
    hipError_t e = hipMemsetAsync(dw1, 0, sizeof(double) * 3 * c, s);
    if (e != hipSuccess) return static_cast<int>(e);
  } else if (!dy) {
    return CODA_EINVAL;
  }
  if (p == 0) return CODA_OK;
  if (!da || !src || !prm) return CODA_EINVAL;
  clear_sticky_error();
  if (w1) hipLaunchKernelGGL(relu_bn_bwd_apply_kernel<true>, dim3(nblocks(p)), dim3(kT), 0, s, da, src, w1, prm, p, c, row_weight, dy, dw1);
  else hipLaunchKernelGGL(relu_bn_bwd_apply_kernel<false>, dim3(nblocks(p)), dim3(kT), 0, s, da, src, w1, prm, p, c, row_weight, dy, dw1);
  return launch_status();
}

CODA_API int coda_sa_bn_finalize_f32(const double *sums, double n, double eps, float momentum, const float *gamma,
                                     const float *beta, float *running_mean, float *running_var,
                                     long long *num_batches, float *stats, int c, void *stream) {
  using namespace coda;
  if (c <= 0 || n <= 0.0 || !sums || !gamma || !beta || !stats) return CODA_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sums, n,
                     eps, momentum, gamma, beta, running_mean, running_var, num_batches, stats, c);
  return launch_status();
}

CODA_API int coda_sa_bn_bwd_coef_f32(const double *sums, double n, const float *gamma, const float *stats, float *coef,
                                     int layout, float *dbeta, float *dgamma, int c, void *stream) {
  using namespace coda;
  if (c <= 0 || !sums || (layout != 0 && layout != 1)) return CODA_EINVAL;
  if (coef && (!gamma || !stats)) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((c + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sums, n,
                     gamma, stats, coef, layout, dbeta, dgamma, c);
  return launch_status();
}

CODA_API int coda_sa_pool_select_f32(const float *ymax, const float *ymin, const int32_t *amax, const int32_t *amin,
                                     const float *stats, float *ysel, int32_t *sel, float *out, long long groups,
                                     int c, void *stream) {
  using namespace coda;
  if (groups < 0 || c <= 0) return CODA_EINVAL;
  if (groups == 0) return CODA_OK;
  if (!ymax || !ymin || !amax || !amin || !stats || !ysel || !sel || !out) return CODA_EINVAL;
  const long long total = groups * c;
  long long blocks = (total + kT * 4 - 1) / (kT * 4);
  blocks = blocks > 8192 ? 8192 : blocks;
  clear_sticky_error();
  hipLaunchKernelGGL(pool_select_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kT), 0, static_cast<hipStream_t>(stream),
                     ymax, ymin, amax, amin, stats, ysel, sel, out, total, c);
  return launch_status();
}

CODA_API int coda_sa_pool_bwd_stats_f32(const float *gout, const float *out, const float *ysel, const float *stats,
                                        float *d, long long groups, int c, double *sums, void *stream) {
  using namespace coda;
  if (groups < 0 || bad_c(c) || !sums) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (groups == 0) return CODA_OK;
  if (!gout || !out || !ysel || !stats || !d) return CODA_EINVAL;
  clear_sticky_error();
  // 64 rows per block (a block per CU at 16 384 groups; nblocks()'s 256 rows left three quarters of the chip idle and the
  // kernel at 41 us for 58 MB): the 2 C double atomics per block stay a few hundred per address
  const long long want = (groups + 63) / 64;
  const int blocks = static_cast<int>(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
  hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(blocks), dim3(kT), 0, s, gout, out, ysel, stats, d, groups, c, sums);
  return launch_status();
}

CODA_API int coda_sa_compact_groups_f32(const float *grouped, const int64_t *cnt, const int64_t *goff, float *x,
                                        float *row_weight, int32_t *goff32, long long groups, int s_len, long long total,
                                        long long rows_padded, void *stream) {
  using namespace coda;
  if (groups < 0 || s_len <= 0 || total < 0 || rows_padded < total || total > groups * s_len) return CODA_EINVAL;
  if (!goff32 || (groups > 0 && (!grouped || !cnt || !goff)) || (rows_padded > 0 && (!x || !row_weight))) return CODA_EINVAL;
  const long long work = groups * s_len + (rows_padded - total);
  clear_sticky_error();
  if (work == 0) {  // no groups: only the terminating offset
    return static_cast<int>(hipMemsetAsync(goff32, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
  }
  hipLaunchKernelGGL(compact_groups_kernel, dim3(static_cast<unsigned>((work + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), grouped, cnt, goff, x, row_weight, goff32, groups, s_len, total,
                     rows_padded);
  return launch_status();
}
