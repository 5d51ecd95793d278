// token_bn.hip -- batch-norm MLP blocks on channels-last tokens (include/coda_token_ops.h,
// part 1).  z (G, R, C): G independent stacks (the prediction heads), R tokens, C channels.
// 256 threads per block, a thread owns 4 consecutive channels (float4), C/4 threads cover a
// row; a block streams `rows_per_block` rows of one group.  grid = (row blocks, G).
#include "coda_token_ops.h"
#include "common.hip.h"
#include "dropout.hip.h"

namespace coda {
namespace {

constexpr int kT = 256;

struct RowMap {
  int tpr, rpb, cq, rsub;
};
__device__ __forceinline__ RowMap row_map(int c) {
  RowMap m;
  m.tpr = c >> 2;
  m.rpb = kT / m.tpr;
  m.cq = threadIdx.x % m.tpr;
  m.rsub = threadIdx.x / m.tpr;
  return m;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 affine(float4 y, float4 a, float4 b) {
  return make_float4(y.x * a.x + b.x, y.y * a.y + b.y, y.z * a.z + b.z, y.w * a.w + b.w);
}

constexpr int kUnroll = 8;  // row loads a thread keeps in flight in the streaming loops

// The block's sums of NACC quantities -> its own slot of the partial-sum array (P, G, NACC, C): plain stores.  (fp64
// atomics on the G*NACC*C result addresses serialise: 512 blocks of the 16 384-token projection spent 50 of their 59 us
// queueing on 512 addresses; the finalize kernels add the P slots in index order instead -- deterministic as well.)
template <int NACC>
__device__ __forceinline__ void block_reduce_to_part(const float4 (&acc)[NACC], const RowMap &m, int c,
                                                     double *part) {
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
      double *dst = part + static_cast<size_t>(k) * c + 4 * m.cq;
      *reinterpret_cast<double2 *>(dst) = make_double2(t0, t1);
      *reinterpret_cast<double2 *>(dst + 2) = make_double2(t2, t3);
    }
  }
}

// sum over the P slots of (P, G, 2, C) for 16 channels of group g: 16 x 16 threads, thread (pl, cl) adds the slots
// pl, pl + 16, ... of channel c0 + cl in index order, lane pl == 0 then adds the 16 sub-sums in order.  Returns the two
// totals to the threads with pl == 0 (others: unspecified).
constexpr int kFinCh = 16, kFinT = 256;
__device__ __forceinline__ void sum_parts(const double *__restrict__ parts, int nparts, int groups, int g, int c, int ch,
                                          bool live, double &t0, double &t1) {
  __shared__ double s_sub[2][kFinT];
  const int pl = threadIdx.x / kFinCh;
  double a0 = 0.0, a1 = 0.0;
  if (live) {
    const size_t stride = static_cast<size_t>(groups) * 2 * c;
    const double *src = parts + (static_cast<size_t>(g) * 2) * c + ch;
    int p = pl;
    for (; p + 3 * (kFinT / kFinCh) < nparts; p += 4 * (kFinT / kFinCh)) {  // 8 loads in flight
      double v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double *q = src + (p + u * (kFinT / kFinCh)) * stride;
        v0[u] = q[0];
        v1[u] = q[c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0 += v0[u]; a1 += v1[u]; }
    }
    for (; p < nparts; p += kFinT / kFinCh) {
      a0 += src[p * stride];
      a1 += src[p * stride + c];
    }
  }
  s_sub[0][threadIdx.x] = a0;
  s_sub[1][threadIdx.x] = a1;
  __syncthreads();
  t0 = t1 = 0.0;
  if (pl == 0) {
    for (int q = 0; q < kFinT / kFinCh; ++q) {
      t0 += s_sub[0][q * kFinCh + threadIdx.x];
      t1 += s_sub[1][q * kFinCh + threadIdx.x];
    }
  }
}

// multiplier of the 4 elements at linear index e0..e0+3: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float4 keep4(uint32_t seed, uint32_t e0, uint32_t thresh24, float inv_keep) {
  if (thresh24 == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
  return make_float4(keep_elem(seed, e0, thresh24) ? inv_keep : 0.f, keep_elem(seed, e0 + 1, thresh24) ? inv_keep : 0.f,
                     keep_elem(seed, e0 + 2, thresh24) ? inv_keep : 0.f,
                     keep_elem(seed, e0 + 3, thresh24) ? inv_keep : 0.f);
}

struct Drop {
  uint32_t thresh24, seed;
  const uint64_t *seed_dev;
  float inv_keep;
};

__global__ __launch_bounds__(kT) void bn_stats_kernel(const float *__restrict__ z, long long rows, int c,
                                                      int rows_per_block, double *__restrict__ parts) {
  const RowMap m = row_map(c);
  const int g = blockIdx.y;
  const float *zg = z + static_cast<size_t>(g) * rows * c;
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_block;
  const long long r1 = min(r0 + rows_per_block, rows);
  auto add = [&](float4 y) {
    acc[0].x += y.x; acc[0].y += y.y; acc[0].z += y.z; acc[0].w += y.w;
    acc[1].x += y.x * y.x; acc[1].y += y.y * y.y; acc[1].z += y.z * y.z; acc[1].w += y.w * y.w;
  };
  // kUnroll row loads in flight per thread (one load per iteration left the kernel latency-bound at 2 TB/s);
  // the accumulation order is unchanged
  long long r = r0 + m.rsub;
  for (; r + (kUnroll - 1) * m.rpb < r1; r += kUnroll * m.rpb) {
    float4 y[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) y[u] = ld4(zg + (r + u * m.rpb) * c + 4 * m.cq);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) add(y[u]);
  }
  for (; r < r1; r += m.rpb) add(ld4(zg + r * c + 4 * m.cq));
  block_reduce_to_part<2>(acc, m, c, parts + (static_cast<size_t>(blockIdx.x) * gridDim.y + g) * 2 * c);
}

__global__ __launch_bounds__(kFinT) void bn_finalize_kernel(const double *__restrict__ sums, int nparts, int groups,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int c, double count,
                                                            float eps, float *__restrict__ prm,
                                                            float *__restrict__ stat) {
  const int g = blockIdx.y;
  const int ch = blockIdx.x * kFinCh + threadIdx.x % kFinCh;
  double t0, t1;
  sum_parts(sums, nparts, groups, g, c, ch, ch < c, t0, t1);
  if (threadIdx.x >= kFinCh || ch >= c) return;
  const double mean = t0 / count;
  double var = t1 / count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  const float meanf = static_cast<float>(mean);
  const float scale = gamma[static_cast<size_t>(g) * c + ch] * invstd;
  float *p = prm + static_cast<size_t>(g) * 4 * c + ch;
  p[0] = scale;
  p[c] = beta[static_cast<size_t>(g) * c + ch] - meanf * scale;
  p[2 * c] = meanf;
  p[3 * c] = invstd;
  if (stat) {
    const double denom = count > 1.0 ? count - 1.0 : 1.0;
    stat[(static_cast<size_t>(g) * 2) * c + ch] = meanf;
    stat[(static_cast<size_t>(g) * 2 + 1) * c + ch] = static_cast<float>(var * (count / denom));
  }
}

template <bool RELU>
__global__ __launch_bounds__(kT) void bn_act_kernel(const float *__restrict__ z, const float *__restrict__ prm,
                                                    long long rows, int c, int rows_per_block, Drop dr,
                                                    float *__restrict__ out) {
  const RowMap m = row_map(c);
  const int g = blockIdx.y;
  const size_t goff = static_cast<size_t>(g) * rows * c;
  const float *pg = prm + static_cast<size_t>(g) * 4 * c + 4 * m.cq;
  const float4 a = ld4(pg), b = ld4(pg + c);
  const uint32_t seed = dr.thresh24 ? fold_seed(dr.seed, dr.seed_dev) : 0u;
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_block;
  const long long r1 = min(r0 + rows_per_block, rows);
  auto one = [&](size_t off, float4 zin) {
    float4 v = affine(zin, a, b);
    if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    const float4 k = keep4(seed, static_cast<uint32_t>(off), dr.thresh24, dr.inv_keep);
    st4(out + off, make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w));
  };
  long long r = r0 + m.rsub;
  for (; r + (kUnroll - 1) * m.rpb < r1; r += kUnroll * m.rpb) {
    float4 y[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) y[u] = ld4(z + goff + (r + u * m.rpb) * c + 4 * m.cq);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) one(goff + (r + u * m.rpb) * c + 4 * m.cq, y[u]);
  }
  for (; r < r1; r += m.rpb) {
    const size_t off = goff + r * c + 4 * m.cq;
    one(off, ld4(z + off));
  }
}

// d = da * keep (* [act > 0])
template <bool RELU>
__device__ __forceinline__ float4 masked_grad(float4 g, float4 y, float4 a, float4 b, float4 k) {
  float4 d = make_float4(g.x * k.x, g.y * k.y, g.z * k.z, g.w * k.w);
  if (RELU) {
    const float4 act = affine(y, a, b);
    d.x = act.x > 0.f ? d.x : 0.f; d.y = act.y > 0.f ? d.y : 0.f;
    d.z = act.z > 0.f ? d.z : 0.f; d.w = act.w > 0.f ? d.w : 0.f;
  }
  return d;
}

template <bool RELU>
__global__ __launch_bounds__(kT) void bn_act_bwd_stats_kernel(const float *__restrict__ da,
                                                              const float *__restrict__ z,
                                                              const float *__restrict__ prm, long long rows,
                                                              int c, int rows_per_block, Drop dr,
                                                              double *__restrict__ parts) {
  const RowMap m = row_map(c);
  const int g = blockIdx.y;
  const size_t goff = static_cast<size_t>(g) * rows * c;
  const float *pg = prm + static_cast<size_t>(g) * 4 * c + 4 * m.cq;
  const float4 a = ld4(pg), b = ld4(pg + c), mu = ld4(pg + 2 * c), is = ld4(pg + 3 * c);
  const uint32_t seed = dr.thresh24 ? fold_seed(dr.seed, dr.seed_dev) : 0u;
  float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_block;
  const long long r1 = min(r0 + rows_per_block, rows);
  auto one = [&](size_t off, float4 y, float4 g_in) {
    const float4 k = keep4(seed, static_cast<uint32_t>(off), dr.thresh24, dr.inv_keep);
    const float4 d = masked_grad<RELU>(g_in, y, a, b, k);
    acc[0].x += d.x; acc[0].y += d.y; acc[0].z += d.z; acc[0].w += d.w;
    acc[1].x += d.x * (y.x - mu.x) * is.x; acc[1].y += d.y * (y.y - mu.y) * is.y;
    acc[1].z += d.z * (y.z - mu.z) * is.z; acc[1].w += d.w * (y.w - mu.w) * is.w;
  };
  constexpr int U2 = kUnroll / 2;  // two input streams
  long long r = r0 + m.rsub;
  for (; r + (U2 - 1) * m.rpb < r1; r += U2 * m.rpb) {
    float4 y[U2], gi[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const size_t off = goff + (r + u * m.rpb) * c + 4 * m.cq;
      y[u] = ld4(z + off);
      gi[u] = ld4(da + off);
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) one(goff + (r + u * m.rpb) * c + 4 * m.cq, y[u], gi[u]);
  }
  for (; r < r1; r += m.rpb) {
    const size_t off = goff + r * c + 4 * m.cq;
    one(off, ld4(z + off), ld4(da + off));
  }
  block_reduce_to_part<2>(acc, m, c, parts + (static_cast<size_t>(blockIdx.x) * gridDim.y + g) * 2 * c);
}

__global__ __launch_bounds__(kFinT) void bn_bwd_finalize_kernel(const double *__restrict__ sums_local, int nparts,
                                                                const double *__restrict__ sums_total, int groups,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ prm, int c, double count,
                                                                float *__restrict__ prmb, float *__restrict__ dgamma,
                                                                float *__restrict__ dbeta) {
  const int g = blockIdx.y;
  const int ch = blockIdx.x * kFinCh + threadIdx.x % kFinCh;
  double l0, l1;
  sum_parts(sums_local, nparts, groups, g, c, ch, ch < c, l0, l1);
  if (threadIdx.x >= kFinCh || ch >= c) return;
  const size_t s0 = (static_cast<size_t>(g) * 2) * c + ch, s1 = s0 + c;
  const double tot0 = sums_total ? sums_total[s0] : l0, tot1 = sums_total ? sums_total[s1] : l1;
  const float invstd = prm[static_cast<size_t>(g) * 4 * c + 3 * c + ch];
  float *p = prmb + static_cast<size_t>(g) * 3 * c + ch;
  p[0] = gamma[static_cast<size_t>(g) * c + ch] * invstd;
  p[c] = static_cast<float>(tot0 / count);
  p[2 * c] = static_cast<float>(tot1 / count);
  dbeta[static_cast<size_t>(g) * c + ch] = static_cast<float>(l0);
  dgamma[static_cast<size_t>(g) * c + ch] = static_cast<float>(l1);
}

template <bool RELU>
__global__ __launch_bounds__(kT) void bn_act_bwd_apply_kernel(const float *__restrict__ da,
                                                              const float *__restrict__ z,
                                                              const float *__restrict__ prm,
                                                              const float *__restrict__ prmb, long long rows,
                                                              int c, int rows_per_block, Drop dr,
                                                              float *__restrict__ dz) {
  const RowMap m = row_map(c);
  const int g = blockIdx.y;
  const size_t goff = static_cast<size_t>(g) * rows * c;
  const float *pg = prm + static_cast<size_t>(g) * 4 * c + 4 * m.cq;
  const float *pb = prmb + static_cast<size_t>(g) * 3 * c + 4 * m.cq;
  const float4 a = ld4(pg), b = ld4(pg + c), mu = ld4(pg + 2 * c), is = ld4(pg + 3 * c);
  const float4 ca = ld4(pb), m1 = ld4(pb + c), m2 = ld4(pb + 2 * c);
  const uint32_t seed = dr.thresh24 ? fold_seed(dr.seed, dr.seed_dev) : 0u;
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_block;
  const long long r1 = min(r0 + rows_per_block, rows);
  auto one = [&](size_t off, float4 y, float4 g_in) {
    const float4 k = keep4(seed, static_cast<uint32_t>(off), dr.thresh24, dr.inv_keep);
    const float4 d = masked_grad<RELU>(g_in, y, a, b, k);
    float4 o;
    o.x = ca.x * (d.x - m1.x - (y.x - mu.x) * is.x * m2.x);
    o.y = ca.y * (d.y - m1.y - (y.y - mu.y) * is.y * m2.y);
    o.z = ca.z * (d.z - m1.z - (y.z - mu.z) * is.z * m2.z);
    o.w = ca.w * (d.w - m1.w - (y.w - mu.w) * is.w * m2.w);
    st4(dz + off, o);
  };
  constexpr int U2 = kUnroll / 2;
  long long r = r0 + m.rsub;
  for (; r + (U2 - 1) * m.rpb < r1; r += U2 * m.rpb) {
    float4 y[U2], gi[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const size_t off = goff + (r + u * m.rpb) * c + 4 * m.cq;
      y[u] = ld4(z + off);
      gi[u] = ld4(da + off);
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) one(goff + (r + u * m.rpb) * c + 4 * m.cq, y[u], gi[u]);
  }
  for (; r < r1; r += m.rpb) {
    const size_t off = goff + r * c + 4 * m.cq;
    one(off, ld4(z + off), ld4(da + off));
  }
}

bool bad_c(int c) { return c < 4 || c > 1024 || (c % 4) != 0 || (kT % (c / 4)) != 0; }

// rows streamed by one block.  The statistics kernels end in 2*C fp64 atomics per block on
// G*2*C addresses, so blocks are kept to ~2 per CU (measured: 2000 blocks of 48 rows ran the
// heads' statistics at 1.2 TB/s, the atomics dominating); the element-wise kernels take the
// same grid, 512 blocks x 256 threads is enough to stream at HBM rate.
int rows_per_block(int groups, long long rows, int c) {
  const int rpb = kT / (c / 4);
  long long want = (static_cast<long long>(groups) * rows + 511) / 512;  // ~512 blocks
  long long n = ((want + rpb - 1) / rpb) * rpb;
  const long long lo = 8LL * rpb, hi = 4096;
  if (n < lo) n = lo;
  if (n > hi) n = hi;
  return static_cast<int>(n);
}
dim3 row_grid(int groups, long long rows, int rpb_rows) {
  return dim3(static_cast<unsigned>((rows + rpb_rows - 1) / rpb_rows), static_cast<unsigned>(groups));
}
bool bad_p(float p) { return !(p >= 0.f) || p >= 1.f; }
Drop make_drop(float p, uint64_t seed, const uint64_t *seed_dev) {
  Drop d;
  d.thresh24 = drop_thresh24(p);
  d.seed = static_cast<uint32_t>(seed ^ (seed >> 32));
  d.seed_dev = seed_dev;
  d.inv_keep = 1.0f / (1.0f - p);
  return d;
}

}  // namespace
}  // namespace coda

using namespace coda;

CODA_API int coda_tok_bn_parts(int groups, long long rows, int c) {
  if (groups <= 0 || rows < 0 || bad_c(c)) return CODA_EINVAL;
  if (rows == 0) return 1;
  return static_cast<int>(row_grid(groups, rows, rows_per_block(groups, rows, c)).x);
}

CODA_API int coda_tok_bn_stats_f32(const float *z, int groups, long long rows, int c, double *parts,
                                   void *stream) {
  if (groups <= 0 || rows < 0 || bad_c(c) || !parts) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) return static_cast<int>(hipMemsetAsync(parts, 0, sizeof(double) * 2 * c * groups, s));
  if (!z) return CODA_EINVAL;
  const int rb = rows_per_block(groups, rows, c);
  clear_sticky_error();
  hipLaunchKernelGGL(bn_stats_kernel, row_grid(groups, rows, rb), dim3(kT), 0, s, z, rows, c, rb, parts);
  return launch_status();
}

CODA_API int coda_tok_bn_finalize_f32(const double *sums, int nparts, const float *gamma, const float *beta,
                                      int groups, int c, double count, float eps, float *prm, float *stat,
                                      void *stream) {
  if (groups <= 0 || c <= 0 || nparts <= 0 || !(count > 0.0) || !sums || !gamma || !beta || !prm) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + kFinCh - 1) / kFinCh, groups), dim3(kFinT), 0,
                     static_cast<hipStream_t>(stream), sums, nparts, groups, gamma, beta, c, count, eps, prm, stat);
  return launch_status();
}

CODA_API int coda_tok_bn_act_f32(const float *z, const float *prm, int groups, long long rows, int c, int relu,
                                 float dropout_p, uint64_t seed, const uint64_t *seed_dev, float *a,
                                 void *stream) {
  if (groups <= 0 || rows < 0 || bad_c(c) || bad_p(dropout_p)) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!z || !prm || !a) return CODA_EINVAL;
  const int rb = rows_per_block(groups, rows, c);
  const Drop dr = make_drop(dropout_p, seed, seed_dev);
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  if (relu) hipLaunchKernelGGL(bn_act_kernel<true>, row_grid(groups, rows, rb), dim3(kT), 0, s, z, prm, rows, c, rb, dr, a);
  else hipLaunchKernelGGL(bn_act_kernel<false>, row_grid(groups, rows, rb), dim3(kT), 0, s, z, prm, rows, c, rb, dr, a);
  return launch_status();
}

CODA_API int coda_tok_bn_act_bwd_stats_f32(const float *da, const float *z, const float *prm, int groups,
                                           long long rows, int c, int relu, float dropout_p, uint64_t seed,
                                           const uint64_t *seed_dev, double *parts, void *stream) {
  if (groups <= 0 || rows < 0 || bad_c(c) || bad_p(dropout_p) || !parts) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) return static_cast<int>(hipMemsetAsync(parts, 0, sizeof(double) * 2 * c * groups, s));
  if (!da || !z || !prm) return CODA_EINVAL;
  const int rb = rows_per_block(groups, rows, c);
  const Drop dr = make_drop(dropout_p, seed, seed_dev);
  clear_sticky_error();
  if (relu) hipLaunchKernelGGL(bn_act_bwd_stats_kernel<true>, row_grid(groups, rows, rb), dim3(kT), 0, s, da, z, prm, rows, c, rb, dr, parts);
  else hipLaunchKernelGGL(bn_act_bwd_stats_kernel<false>, row_grid(groups, rows, rb), dim3(kT), 0, s, da, z, prm, rows, c, rb, dr, parts);
  return launch_status();
}

CODA_API int coda_tok_bn_bwd_finalize_f32(const double *sums_local, int nparts, const double *sums_total,
                                          const float *gamma, const float *prm, int groups, int c, double count,
                                          float *prmb, float *dgamma, float *dbeta, void *stream) {
  if (groups <= 0 || c <= 0 || nparts <= 0 || !(count > 0.0) || !sums_local || !gamma || !prm || !prmb || !dgamma ||
      !dbeta)
    return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + kFinCh - 1) / kFinCh, groups), dim3(kFinT), 0,
                     static_cast<hipStream_t>(stream), sums_local, nparts, sums_total, groups, gamma, prm, c, count,
                     prmb, dgamma, dbeta);
  return launch_status();
}

CODA_API int coda_tok_bn_act_bwd_apply_f32(const float *da, const float *z, const float *prm, const float *prmb,
                                           int groups, long long rows, int c, int relu, float dropout_p,
                                           uint64_t seed, const uint64_t *seed_dev, float *dz, void *stream) {
  if (groups <= 0 || rows < 0 || bad_c(c) || bad_p(dropout_p)) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!da || !z || !prm || !prmb || !dz) return CODA_EINVAL;
  const int rb = rows_per_block(groups, rows, c);
  const Drop dr = make_drop(dropout_p, seed, seed_dev);
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  if (relu) hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, row_grid(groups, rows, rb), dim3(kT), 0, s, da, z, prm, prmb, rows, c, rb, dr, dz);
  else hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, row_grid(groups, rows, rb), dim3(kT), 0, s, da, z, prm, prmb, rows, c, rb, dr, dz);
  return launch_status();
}
