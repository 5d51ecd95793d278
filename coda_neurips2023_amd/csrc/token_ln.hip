// token_ln.hip -- glue of the pre-norm transformer layers (include/coda_token_ops.h, part 2):
// bias + dropout + residual + LayerNorm + positional add in one pass, forward and backward,
// and the feed-forward bias + ReLU + dropout.
//
// LayerNorm kernels: one wave per token row, a lane owns float4 chunks at channels
// 256*k + 4*lane (k < NV), so each wave instruction reads/writes 1 KiB contiguous.  Row
// statistics are wave reductions (two-pass: mean, then centred variance).  A forward block
// is 4 waves; the backward keeps per-lane column accumulators over the rows its wave visits
// and writes one (3,C) partial per block, so its blocks are 16 / NV waves (48 KB of LDS for
// the cross-wave sum): the same number of partials with 4 x the waves in flight -- the
// decoder's 2048 rows are one row per wave, the encoder's 16 384 four.
#include "coda_token_ops.h"
#include "common.hip.h"
#include "dropout.hip.h"

namespace coda {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * kWave;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

struct Drop {
  uint32_t thresh24, seed;
  const uint64_t *seed_dev;
  float inv_keep;
};
__device__ __forceinline__ float4 drop4(float4 v, uint32_t seed, uint32_t e0, const Drop &dr) {
  return make_float4(keep_elem(seed, e0, dr.thresh24) ? v.x * dr.inv_keep : 0.f,
                     keep_elem(seed, e0 + 1, dr.thresh24) ? v.y * dr.inv_keep : 0.f,
                     keep_elem(seed, e0 + 2, dr.thresh24) ? v.z * dr.inv_keep : 0.f,
                     keep_elem(seed, e0 + 3, dr.thresh24) ? v.w * dr.inv_keep : 0.f);
}

struct LnFwd {
  const float *x, *bias, *res, *pos, *gamma, *beta;
  float *s_out, *y_out, *yp_out, *mean, *rstd;
  long long rows;
  int c;
  float eps;
  Drop dr;
  // second head (coda_tok_add_ln_fwd2_f32): y2 = LayerNorm(s) * gamma2 + beta2 of the SAME s (same mean / rstd), yp2 = y2 + pos2
  const float *gamma2 = nullptr, *beta2 = nullptr, *pos2 = nullptr;
  float *y2_out = nullptr, *yp2_out = nullptr;
};

template <int NV>
__global__ __launch_bounds__(kThreads) void add_ln_fwd_kernel(LnFwd p) {
  const int lane = lane_id(), w = wave_id();
  const uint32_t seed = p.dr.thresh24 ? fold_seed(p.dr.seed, p.dr.seed_dev) : 0u;
  const float inv_c = 1.0f / static_cast<float>(p.c);
  bool on[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) on[k] = 256 * k + 4 * lane < p.c;
  for (long long r = static_cast<long long>(blockIdx.x) * kWavesPerBlock + w; r < p.rows;
       r += static_cast<long long>(gridDim.x) * kWavesPerBlock) {
    const size_t base = static_cast<size_t>(r) * p.c;
    float4 s[NV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on[k]) {
        const int ch = 256 * k + 4 * lane;
        float4 v = ld4(p.x + base + ch);
        if (p.bias) v = add4(v, ld4(p.bias + ch));
        if (p.dr.thresh24) v = drop4(v, seed, static_cast<uint32_t>(base + ch), p.dr);
        if (p.res) v = add4(v, ld4(p.res + base + ch));
        s[k] = v;
        if (p.s_out) st4(p.s_out + base + ch, v);
        sum += (v.x + v.y) + (v.z + v.w);
      }
    }
    if (!p.gamma) continue;
    const float mu = wave_sum(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (on[k]) {
        const float a = s[k].x - mu, b = s[k].y - mu, c2 = s[k].z - mu, d = s[k].w - mu;
        sq += (a * a + b * b) + (c2 * c2 + d * d);
      }
    const float rs = rsqrtf(wave_sum(sq) * inv_c + p.eps);
    if (lane == 0) {
      p.mean[r] = mu;
      p.rstd[r] = rs;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (on[k]) {
        const int ch = 256 * k + 4 * lane;
        const float4 g = ld4(p.gamma + ch), b = ld4(p.beta + ch);
        float4 y;
        y.x = (s[k].x - mu) * rs * g.x + b.x;
        y.y = (s[k].y - mu) * rs * g.y + b.y;
        y.z = (s[k].z - mu) * rs * g.z + b.z;
        y.w = (s[k].w - mu) * rs * g.w + b.w;
        st4(p.y_out + base + ch, y);
        if (p.yp_out) st4(p.yp_out + base + ch, add4(y, ld4(p.pos + base + ch)));
        if (p.gamma2) {
          const float4 g2 = ld4(p.gamma2 + ch), b2 = ld4(p.beta2 + ch);
          float4 y2;
          y2.x = (s[k].x - mu) * rs * g2.x + b2.x;
          y2.y = (s[k].y - mu) * rs * g2.y + b2.y;
          y2.z = (s[k].z - mu) * rs * g2.z + b2.z;
          y2.w = (s[k].w - mu) * rs * g2.w + b2.w;
          st4(p.y2_out + base + ch, y2);
          if (p.yp2_out) st4(p.yp2_out + base + ch, add4(y2, ld4(p.pos2 + base + ch)));
        }
      }
  }
}

struct LnBwd {
  const float *dy, *dyp, *ds, *s, *mean, *rstd, *gamma;
  float *dres_out, *dx_out, *partials;
  long long rows;
  int c;
  Drop dr;
  // second head (coda_tok_add_ln_bwd2_f32): a second LayerNorm of the same s with its own affine map and upstream
  // gradient dy2 (+ dyp2); its [sum d2 * xhat | sum d2 | 0] partials go to partials2
  const float *dy2 = nullptr, *dyp2 = nullptr, *gamma2 = nullptr;
  float *partials2 = nullptr;
  // positional-embedding gradient folded in: dpos_acc (+)= dpos_src + dpos_extra (dpos_src = this call's dyp or dyp2)
  const float *dpos_src = nullptr, *dpos_extra = nullptr;
  float *dpos_acc = nullptr;
  int dpos_init = 0;
};

constexpr int bwd_waves(int nv) { return 16 / nv; }

// DUAL: the second head of LnBwd (two LayerNorms of one s: the decoder's layer-output norm and the next layer's norm1)
template <int NV, bool DUAL = false>
__global__ __launch_bounds__(bwd_waves(NV) * kWave) void add_ln_bwd_kernel(LnBwd p) {
  constexpr int kWavesPerBlock = bwd_waves(NV);  // (shadows the forward's 4)
  __shared__ float4 s_acc[kWavesPerBlock][3][NV][kWave];
  const int lane = lane_id(), w = wave_id();
  const uint32_t seed = p.dr.thresh24 ? fold_seed(p.dr.seed, p.dr.seed_dev) : 0u;
  const float inv_c = 1.0f / static_cast<float>(p.c);
  const bool has_ln = p.gamma != nullptr;
  bool on[NV];
  float4 gam[NV], a_g[NV], a_b[NV], a_x[NV];
  float4 gam2[DUAL ? NV : 1], a_g2[DUAL ? NV : 1], a_b2[DUAL ? NV : 1];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    on[k] = 256 * k + 4 * lane < p.c;
    gam[k] = (has_ln && on[k]) ? ld4(p.gamma + 256 * k + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    a_g[k] = a_b[k] = a_x[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DUAL) {
      gam2[k] = on[k] ? ld4(p.gamma2 + 256 * k + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
      a_g2[k] = a_b2[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (long long r = static_cast<long long>(blockIdx.x) * kWavesPerBlock + w; r < p.rows;
       r += static_cast<long long>(gridDim.x) * kWavesPerBlock) {
    const size_t base = static_cast<size_t>(r) * p.c;
    float4 g[NV], xh[NV];
    float c1 = 0.f, c2 = 0.f, mu = 0.f, rs = 0.f;
    if (has_ln) {
      mu = p.mean[r];
      rs = p.rstd[r];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        g[k] = xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on[k]) {
          const int ch = 256 * k + 4 * lane;
          float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.dy) d = ld4(p.dy + base + ch);
          if (p.dyp) d = add4(d, ld4(p.dyp + base + ch));
          const float4 sv = ld4(p.s + base + ch);
          xh[k] = make_float4((sv.x - mu) * rs, (sv.y - mu) * rs, (sv.z - mu) * rs, (sv.w - mu) * rs);
          a_b[k] = add4(a_b[k], d);
          a_g[k].x += d.x * xh[k].x; a_g[k].y += d.y * xh[k].y; a_g[k].z += d.z * xh[k].z; a_g[k].w += d.w * xh[k].w;
          g[k] = make_float4(d.x * gam[k].x, d.y * gam[k].y, d.z * gam[k].z, d.w * gam[k].w);
          if (DUAL) {
            float4 d2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.dy2) d2 = ld4(p.dy2 + base + ch);
            if (p.dyp2) d2 = add4(d2, ld4(p.dyp2 + base + ch));
            a_b2[k] = add4(a_b2[k], d2);
            a_g2[k].x += d2.x * xh[k].x; a_g2[k].y += d2.y * xh[k].y; a_g2[k].z += d2.z * xh[k].z; a_g2[k].w += d2.w * xh[k].w;
            g[k].x += d2.x * gam2[k].x; g[k].y += d2.y * gam2[k].y; g[k].z += d2.z * gam2[k].z; g[k].w += d2.w * gam2[k].w;
          }
          if (p.dpos_acc) {  // the positional embedding's gradient: what came in through y + pos (+ a second such tensor)
            float4 e = ld4(p.dpos_src + base + ch);
            if (p.dpos_extra) e = add4(e, ld4(p.dpos_extra + base + ch));
            if (!p.dpos_init) e = add4(e, ld4(p.dpos_acc + base + ch));
            st4(p.dpos_acc + base + ch, e);
          }
          c1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
          c2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
        }
      }
      c1 = wave_sum(c1) * inv_c;
      c2 = wave_sum(c2) * inv_c;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (on[k]) {
        const int ch = 256 * k + 4 * lane;
        float4 dres = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_ln) {
          dres.x = rs * (g[k].x - c1 - xh[k].x * c2);
          dres.y = rs * (g[k].y - c1 - xh[k].y * c2);
          dres.z = rs * (g[k].z - c1 - xh[k].z * c2);
          dres.w = rs * (g[k].w - c1 - xh[k].w * c2);
        }
        if (p.ds) dres = add4(dres, ld4(p.ds + base + ch));
        if (p.dres_out) st4(p.dres_out + base + ch, dres);
        float4 dx = dres;
        if (p.dr.thresh24) dx = drop4(dres, seed, static_cast<uint32_t>(base + ch), p.dr);
        if (p.dx_out) st4(p.dx_out + base + ch, dx);
        a_x[k] = add4(a_x[k], dx);
      }
  }
  // block partial: sum the 4 waves' column accumulators
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    s_acc[w][0][k][lane] = a_g[k];
    s_acc[w][1][k][lane] = a_b[k];
    s_acc[w][2][k][lane] = a_x[k];
  }
  __syncthreads();
  if (w < 3) {  // wave j reduces accumulator j
    float *dst = p.partials + (static_cast<size_t>(blockIdx.x) * 3 + w) * p.c;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (on[k]) {
        float4 t = s_acc[0][w][k][lane];
#pragma unroll
        for (int q = 1; q < kWavesPerBlock; ++q) t = add4(t, s_acc[q][w][k][lane]);
        st4(dst + 256 * k + 4 * lane, t);
      }
  }
  if (DUAL) {  // the second head's two accumulators through the same LDS; its third partial row is zero
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      s_acc[w][0][k][lane] = a_g2[k];
      s_acc[w][1][k][lane] = a_b2[k];
    }
    __syncthreads();
    if (w < 3) {
      float *dst = p.partials2 + (static_cast<size_t>(blockIdx.x) * 3 + w) * p.c;
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (on[k]) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (w < 2) {
            t = s_acc[0][w][k][lane];
#pragma unroll
            for (int q = 1; q < kWavesPerBlock; ++q) t = add4(t, s_acc[q][w][k][lane]);
          }
          st4(dst + 256 * k + 4 * lane, t);
        }
    }
  }
}

// out[g][j] = sum_b partials[g][b][j]: 64 columns x 16 block-slices per workgroup, fixed order
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float *__restrict__ partials, int blocks,
                                                               int n, float *__restrict__ out) {
  __shared__ float s_part[16][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const float *pg = partials + static_cast<size_t>(blockIdx.y) * blocks * n;
  float t = 0.f;
  if (col < n) {
    int b = slice;
    for (; b + 48 < blocks; b += 64) {  // four independent loads in flight
      const float v0 = pg[static_cast<size_t>(b) * n + col], v1 = pg[static_cast<size_t>(b + 16) * n + col];
      const float v2 = pg[static_cast<size_t>(b + 32) * n + col], v3 = pg[static_cast<size_t>(b + 48) * n + col];
      t += (v0 + v1) + (v2 + v3);
    }
    for (; b < blocks; b += 16) t += pg[static_cast<size_t>(b) * n + col];
  }
  s_part[slice][lane] = t;
  __syncthreads();
  if (slice == 0 && col < n) {
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += s_part[q][lane];
    out[static_cast<size_t>(blockIdx.y) * n + col] = r;
  }
}

constexpr int kMaxColsumItems = 120;  // 24 B each: 2.9 KB of kernel arguments
struct ColsumBatch {
  CodaColsumItem item[kMaxColsumItems];
};

// Items of at most 16 partial rows (the split-K chunk sums of the weight gradients: 8 rows x up to 196 608 columns) take
// four columns per thread: a quarter of the workgroups, 16-byte accesses, the same order of additions.
__host__ __device__ __forceinline__ bool colsum_item_wide(const CodaColsumItem &it) {
  return it.blocks <= 16 && (it.n & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(it.partials) | reinterpret_cast<uintptr_t>(it.out)) & 15) == 0;
}
__host__ __device__ __forceinline__ int colsum_item_tiles(const CodaColsumItem &it) {
  return colsum_item_wide(it) ? (it.n + 255) / 256 : (it.n + 63) / 64;
}

// the same reduction for many (partials, out) pairs: blockIdx.y = item, blockIdx.z = group, blockIdx.x = 64 columns
// (256 for the wide items)
__global__ __launch_bounds__(1024) void colsum_finalize_grouped_kernel(const ColsumBatch batch) {
  __shared__ float s_part[16][64];
  __shared__ float4 s_part4[16][64];
  const CodaColsumItem &it = batch.item[blockIdx.y];
  if (static_cast<int>(blockIdx.x) >= colsum_item_tiles(it) || static_cast<int>(blockIdx.z) >= it.groups)
    return;  // block-uniform
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  if (colsum_item_wide(it)) {
    const int col = blockIdx.x * 256 + 4 * lane, n = it.n;
    const float *pg = it.partials + static_cast<size_t>(blockIdx.z) * it.blocks * n;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n && slice < it.blocks) t = ld4(pg + static_cast<size_t>(slice) * n + col);
    s_part4[slice][lane] = t;
    __syncthreads();
    if (slice == 0 && col < n) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 16; ++q) r = add4(r, s_part4[q][lane]);
      st4(it.out + static_cast<size_t>(blockIdx.z) * n + col, r);
    }
    return;
  }
  const int col = blockIdx.x * 64 + lane, n = it.n, blocks = it.blocks;
  const float *pg = it.partials + static_cast<size_t>(blockIdx.z) * blocks * n;
  float t = 0.f;
  if (col < n) {
    int b = slice;
    for (; b + 48 < blocks; b += 64) {
      const float v0 = pg[static_cast<size_t>(b) * n + col], v1 = pg[static_cast<size_t>(b + 16) * n + col];
      const float v2 = pg[static_cast<size_t>(b + 32) * n + col], v3 = pg[static_cast<size_t>(b + 48) * n + col];
      t += (v0 + v1) + (v2 + v3);
    }
    for (; b < blocks; b += 16) t += pg[static_cast<size_t>(b) * n + col];
  }
  s_part[slice][lane] = t;
  __syncthreads();
  if (slice == 0 && col < n) {
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += s_part[q][lane];
    it.out[static_cast<size_t>(blockIdx.z) * n + col] = r;
  }
}

// per-block column sums of x (G, rows, C): partials (G, blocks, C)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, long long rows, int c,
                                                             float *__restrict__ partials) {
  __shared__ float4 s_part[256];
  // blockDim.x = tpr * rpb <= 256 (any c / 4 <= 256: c = 768 runs 192-thread blocks)
  const int tpr = c >> 2, rpb = static_cast<int>(blockDim.x) / tpr, cq = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  const float *xg = x + static_cast<size_t>(blockIdx.y) * rows * c;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  long long r = static_cast<long long>(blockIdx.x) * rpb + rsub;
  const long long step = static_cast<long long>(gridDim.x) * rpb;
  for (; r + step < rows; r += 2 * step) {  // two independent loads in flight
    a0 = add4(a0, ld4(xg + static_cast<size_t>(r) * c + 4 * cq));
    a1 = add4(a1, ld4(xg + static_cast<size_t>(r + step) * c + 4 * cq));
  }
  if (r < rows) a0 = add4(a0, ld4(xg + static_cast<size_t>(r) * c + 4 * cq));
  s_part[threadIdx.x] = add4(a0, a1);
  __syncthreads();
  if (rsub == 0) {
    float4 t = s_part[cq];
    for (int q = 1; q < rpb; ++q) t = add4(t, s_part[q * tpr + cq]);
    st4(partials + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * c + 4 * cq, t);
  }
}

// ---- feed-forward activation ------------------------------------------------------------
constexpr int kT = 256;

__global__ __launch_bounds__(kT) void bias_relu_dropout_fwd_kernel(const float *__restrict__ h,
                                                                   const float *__restrict__ bias, long long rows,
                                                                   int c, Drop dr, float *__restrict__ a) {
  const int tpr = c >> 2, rpb = kT / tpr, cq = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  const uint32_t seed = dr.thresh24 ? fold_seed(dr.seed, dr.seed_dev) : 0u;
  const float4 b = bias ? ld4(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long r = static_cast<long long>(blockIdx.x) * rpb + rsub; r < rows;
       r += static_cast<long long>(gridDim.x) * rpb) {
    const size_t off = static_cast<size_t>(r) * c + 4 * cq;
    float4 v = add4(ld4(h + off), b);
    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (dr.thresh24) v = drop4(v, seed, static_cast<uint32_t>(off), dr);
    st4(a + off, v);
  }
}

__global__ __launch_bounds__(kT) void bias_relu_dropout_bwd_kernel(const float *__restrict__ da,
                                                                   const float *__restrict__ a, long long rows,
                                                                   int c, float inv_keep, float *__restrict__ dz,
                                                                   float *__restrict__ partials) {
  __shared__ float4 s_part[kT];
  const int tpr = c >> 2, rpb = kT / tpr, cq = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long r = static_cast<long long>(blockIdx.x) * rpb + rsub; r < rows;
       r += static_cast<long long>(gridDim.x) * rpb) {
    const size_t off = static_cast<size_t>(r) * c + 4 * cq;
    const float4 g = ld4(da + off), av = ld4(a + off);
    const float4 d = make_float4(av.x > 0.f ? g.x * inv_keep : 0.f, av.y > 0.f ? g.y * inv_keep : 0.f,
                                 av.z > 0.f ? g.z * inv_keep : 0.f, av.w > 0.f ? g.w * inv_keep : 0.f);
    st4(dz + off, d);
    acc = add4(acc, d);
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  if (rsub == 0) {
    float4 t = acc;
    for (int q = 1; q < rpb; ++q) t = add4(t, s_part[q * tpr + cq]);
    st4(partials + static_cast<size_t>(blockIdx.x) * c + 4 * cq, t);
  }
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
bool bad_ln_c(int c) { return c < 4 || c > 1024 || (c % 4) != 0; }
int ln_blocks(long long rows) {  // forward: a wave per row, 4 rows per block and pass
  const long long want = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
  return static_cast<int>(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}
int ln_bwd_nv(int c) { return c <= 256 ? 1 : (c <= 512 ? 2 : 4); }
int ln_bwd_blocks(long long rows, int c) {  // backward: one (3,C) partial per block, at most 256 of them
  const int waves = bwd_waves(ln_bwd_nv(c));
  const long long want = (rows + waves - 1) / waves;
  return static_cast<int>(want < 1 ? 1 : (want > 256 ? 256 : want));
}
bool bad_row_c(int c) { return c < 4 || c > 1024 || (c % 4) != 0 || (kT % (c / 4)) != 0; }
bool bad_colsum_c(int c) { return c < 4 || c > 1024 || (c % 4) != 0; }
int row_blocks(long long rows, int c) {
  const int rpb = kT / (c / 4);
  const long long want = (rows + 4 * rpb - 1) / (4 * rpb);  // >= 4 passes per block
  return static_cast<int>(want < 1 ? 1 : (want > 512 ? 512 : want));
}

}  // namespace
}  // namespace coda

using namespace coda;

CODA_API int coda_tok_add_ln_fwd_f32(const float *x, const float *bias, const float *res, const float *pos,
                                     const float *gamma, const float *beta, long long rows, int c, float eps,
                                     float dropout_p, uint64_t seed, const uint64_t *seed_dev, float *s_out,
                                     float *y_out, float *yp_out, float *mean, float *rstd, void *stream) {
  if (rows < 0 || bad_ln_c(c) || bad_p(dropout_p)) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!x) return CODA_EINVAL;
  const bool changes = bias || res || dropout_p > 0.f;
  if (changes && !s_out) return CODA_EINVAL;
  if (gamma && (!beta || !y_out || !mean || !rstd)) return CODA_EINVAL;
  if (!gamma && !changes) return CODA_EINVAL;  // nothing to do
  if ((pos != nullptr) != (yp_out != nullptr) || (pos && !gamma)) return CODA_EINVAL;
  LnFwd p{x, bias, res, pos, gamma, beta, s_out, y_out, yp_out, mean, rstd, rows, c, eps,
          make_drop(dropout_p, seed, seed_dev)};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(ln_blocks(rows));
  clear_sticky_error();
  if (c <= 256) hipLaunchKernelGGL(add_ln_fwd_kernel<1>, grid, dim3(kThreads), 0, s, p);
  else if (c <= 512) hipLaunchKernelGGL(add_ln_fwd_kernel<2>, grid, dim3(kThreads), 0, s, p);
  else hipLaunchKernelGGL(add_ln_fwd_kernel<4>, grid, dim3(kThreads), 0, s, p);
  return launch_status();
}

CODA_API int coda_tok_add_ln_bwd_blocks(long long rows, int c) {
  if (rows < 0 || bad_ln_c(c)) return CODA_EINVAL;
  return ln_bwd_blocks(rows, c);
}

CODA_API int coda_tok_add_ln_bwd_f32(const float *dy, const float *dyp, const float *ds, const float *s_in,
                                     const float *mean, const float *rstd, const float *gamma, long long rows,
                                     int c, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                                     float *dres_out, float *dx_out, float *partials, float *sums_out,
                                     void *stream) {
  if (rows < 0 || bad_ln_c(c) || bad_p(dropout_p) || !partials) return CODA_EINVAL;
  if (gamma && (!s_in || !mean || !rstd || (!dy && !dyp))) return CODA_EINVAL;
  if (!gamma && !ds) return CODA_EINVAL;
  if (!dres_out && !dx_out) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = ln_bwd_blocks(rows, c);
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(partials, 0, sizeof(float) * 3 * c * blocks, s);
    if (e == hipSuccess && sums_out) e = hipMemsetAsync(sums_out, 0, sizeof(float) * 3 * c, s);
    return e == hipSuccess ? CODA_OK : static_cast<int>(e);
  }
  LnBwd p{dy, dyp, ds, s_in, mean, rstd, gamma, dres_out, dx_out, partials, rows, c,
          make_drop(dropout_p, seed, seed_dev)};
  const dim3 grid(blocks);
  clear_sticky_error();
  if (c <= 256) hipLaunchKernelGGL(add_ln_bwd_kernel<1>, grid, dim3(bwd_waves(1) * kWave), 0, s, p);
  else if (c <= 512) hipLaunchKernelGGL(add_ln_bwd_kernel<2>, grid, dim3(bwd_waves(2) * kWave), 0, s, p);
  else hipLaunchKernelGGL(add_ln_bwd_kernel<4>, grid, dim3(bwd_waves(4) * kWave), 0, s, p);
  if (sums_out)  // second launch of the same call: fixed-order reduction of the per-block partials
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((3 * c + 63) / 64, 1), dim3(1024), 0, s, partials, blocks, 3 * c,
                       sums_out);
  return launch_status();
}

CODA_API int coda_tok_add_ln_fwd2_f32(const float *x, const float *bias, const float *res, const float *pos,
                                      const float *gamma, const float *beta, const float *gamma2, const float *beta2,
                                      const float *pos2, long long rows, int c, float eps, float dropout_p, uint64_t seed,
                                      const uint64_t *seed_dev, float *s_out, float *y_out, float *yp_out, float *y2_out,
                                      float *yp2_out, float *mean, float *rstd, void *stream) {
  if (rows < 0 || bad_ln_c(c) || bad_p(dropout_p)) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!x || !gamma || !beta || !y_out || !mean || !rstd || !gamma2 || !beta2 || !y2_out) return CODA_EINVAL;
  const bool changes = bias || res || dropout_p > 0.f;
  if (changes && !s_out) return CODA_EINVAL;
  if ((pos != nullptr) != (yp_out != nullptr) || (pos2 != nullptr) != (yp2_out != nullptr)) return CODA_EINVAL;
  LnFwd p{x, bias, res, pos, gamma, beta, s_out, y_out, yp_out, mean, rstd, rows, c, eps,
          make_drop(dropout_p, seed, seed_dev)};
  p.gamma2 = gamma2; p.beta2 = beta2; p.pos2 = pos2; p.y2_out = y2_out; p.yp2_out = yp2_out;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(ln_blocks(rows));
  clear_sticky_error();
  if (c <= 256) hipLaunchKernelGGL(add_ln_fwd_kernel<1>, grid, dim3(kThreads), 0, s, p);
  else if (c <= 512) hipLaunchKernelGGL(add_ln_fwd_kernel<2>, grid, dim3(kThreads), 0, s, p);
  else hipLaunchKernelGGL(add_ln_fwd_kernel<4>, grid, dim3(kThreads), 0, s, p);
  return launch_status();
}

CODA_API int coda_tok_add_ln_bwd2_f32(const float *dy, const float *dyp, const float *dy2, const float *dyp2,
                                      const float *ds, const float *s_in, const float *mean, const float *rstd,
                                      const float *gamma, const float *gamma2, long long rows, int c, float dropout_p,
                                      uint64_t seed, const uint64_t *seed_dev, int dpos_from, const float *dpos_extra,
                                      float *dpos_acc, int dpos_init, float *dres_out, float *dx_out, float *partials,
                                      float *partials2, void *stream) {
  if (rows < 0 || bad_ln_c(c) || bad_p(dropout_p) || !partials || !gamma) return CODA_EINVAL;
  if (!s_in || !mean || !rstd || (!dy && !dyp)) return CODA_EINVAL;
  if (gamma2 && ((!dy2 && !dyp2) || !partials2)) return CODA_EINVAL;
  if (!gamma2 && (dy2 || dyp2)) return CODA_EINVAL;
  if (!dres_out && !dx_out) return CODA_EINVAL;
  if (dpos_from < 0 || dpos_from > 2 || (dpos_from == 0) != (dpos_acc == nullptr)) return CODA_EINVAL;
  const float *dpos_src = dpos_from == 1 ? dyp : (dpos_from == 2 ? dyp2 : nullptr);
  if (dpos_from != 0 && !dpos_src) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = ln_bwd_blocks(rows, c);
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(partials, 0, sizeof(float) * 3 * c * blocks, s);
    if (e == hipSuccess && partials2) e = hipMemsetAsync(partials2, 0, sizeof(float) * 3 * c * blocks, s);
    return e == hipSuccess ? CODA_OK : static_cast<int>(e);
  }
  LnBwd p{dy, dyp, ds, s_in, mean, rstd, gamma, dres_out, dx_out, partials, rows, c,
          make_drop(dropout_p, seed, seed_dev)};
  p.dy2 = dy2; p.dyp2 = dyp2; p.gamma2 = gamma2; p.partials2 = partials2;
  p.dpos_src = dpos_src; p.dpos_extra = dpos_extra; p.dpos_acc = dpos_acc; p.dpos_init = dpos_init;
  const dim3 grid(blocks);
  clear_sticky_error();
  if (gamma2) {
    if (c <= 256) hipLaunchKernelGGL((add_ln_bwd_kernel<1, true>), grid, dim3(bwd_waves(1) * kWave), 0, s, p);
    else if (c <= 512) hipLaunchKernelGGL((add_ln_bwd_kernel<2, true>), grid, dim3(bwd_waves(2) * kWave), 0, s, p);
    else hipLaunchKernelGGL((add_ln_bwd_kernel<4, true>), grid, dim3(bwd_waves(4) * kWave), 0, s, p);
  } else {
    if (c <= 256) hipLaunchKernelGGL(add_ln_bwd_kernel<1>, grid, dim3(bwd_waves(1) * kWave), 0, s, p);
    else if (c <= 512) hipLaunchKernelGGL(add_ln_bwd_kernel<2>, grid, dim3(bwd_waves(2) * kWave), 0, s, p);
    else hipLaunchKernelGGL(add_ln_bwd_kernel<4>, grid, dim3(bwd_waves(4) * kWave), 0, s, p);
  }
  return launch_status();
}

CODA_API int coda_tok_colsum_finalize_f32(const float *partials, int blocks, int n, float *out, void *stream) {
  if (blocks <= 0 || n <= 0 || !partials || !out) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((n + 63) / 64, 1), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), partials, blocks, n, out);
  return launch_status();
}

CODA_API int coda_tok_colsum_finalize_grouped_f32(const CodaColsumItem *items, int count, void *stream) {
  if (count < 0) return CODA_EINVAL;
  if (count == 0) return CODA_OK;
  if (!items) return CODA_EINVAL;
  for (int i = 0; i < count; ++i)
    if (!items[i].partials || !items[i].out || items[i].blocks <= 0 || items[i].n <= 0 || items[i].groups <= 0)
      return CODA_EINVAL;
  clear_sticky_error();
  for (int first = 0; first < count; first += kMaxColsumItems) {
    ColsumBatch b;
    const int m = min(kMaxColsumItems, count - first);
    int max_tiles = 0, max_g = 0;
    for (int i = 0; i < m; ++i) {
      b.item[i] = items[first + i];
      max_tiles = max(max_tiles, colsum_item_tiles(b.item[i]));
      max_g = max(max_g, b.item[i].groups);
    }
    hipLaunchKernelGGL(colsum_finalize_grouped_kernel, dim3(max_tiles, m, max_g), dim3(1024), 0,
                       static_cast<hipStream_t>(stream), b);
    const int st = launch_status();
    if (st != CODA_OK) return st;
  }
  return CODA_OK;
}

CODA_API int coda_tok_colsum_blocks(long long rows, int c) {
  if (rows < 0 || bad_colsum_c(c)) return CODA_EINVAL;
  return row_blocks(rows, c);
}

CODA_API int coda_tok_colsum_f32(const float *x, int groups, long long rows, int c, float *partials, float *out,
                                 void *stream) {
  if (groups <= 0 || rows < 0 || bad_colsum_c(c)) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) {
    if (!out) return CODA_EINVAL;  // nothing to defer
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * groups * c, s);
    return e == hipSuccess ? CODA_OK : static_cast<int>(e);
  }
  if (!x || !partials) return CODA_EINVAL;
  const int blocks = row_blocks(rows, c);
  clear_sticky_error();
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks, groups), dim3((c / 4) * (256 / (c / 4))), 0, s, x, rows, c, partials);
  if (out)
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((c + 63) / 64, groups), dim3(1024), 0, s, partials, blocks, c, out);
  return launch_status();
}

CODA_API int coda_tok_bias_relu_dropout_fwd_f32(const float *h, const float *bias, long long rows, int c,
                                                float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                                                float *a, void *stream) {
  if (rows < 0 || bad_row_c(c) || bad_p(dropout_p)) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!h || !a) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(bias_relu_dropout_fwd_kernel, dim3(row_blocks(rows, c)), dim3(kT), 0,
                     static_cast<hipStream_t>(stream), h, bias, rows, c, make_drop(dropout_p, seed, seed_dev), a);
  return launch_status();
}

CODA_API int coda_tok_bias_relu_dropout_bwd_blocks(long long rows, int c) {
  if (rows < 0 || bad_row_c(c)) return CODA_EINVAL;
  return row_blocks(rows, c);
}

CODA_API int coda_tok_bias_relu_dropout_bwd_f32(const float *da, const float *a, long long rows, int c,
                                                float dropout_p, float *dz, float *partials, float *dbias,
                                                void *stream) {
  if (rows < 0 || bad_row_c(c) || bad_p(dropout_p) || !partials) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = row_blocks(rows, c);
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(partials, 0, sizeof(float) * c * blocks, s);
    if (e == hipSuccess && dbias) e = hipMemsetAsync(dbias, 0, sizeof(float) * c, s);
    return e == hipSuccess ? CODA_OK : static_cast<int>(e);
  }
  if (!da || !a || !dz) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(bias_relu_dropout_bwd_kernel, dim3(blocks), dim3(kT), 0, s, da, a, rows, c,
                     1.0f / (1.0f - dropout_p), dz, partials);
  if (dbias)
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((c + 63) / 64, 1), dim3(1024), 0, s, partials, blocks, c, dbias);
  return launch_status();
}
