// align_loss.hip -- CLIP-space alignment losses of all decoder layers in one pass each way
// (include/coda_align_loss.h).  One wave per proposal row: lane i holds channels
// 64*k + i (k < E/64), so every global access of a wave instruction is a contiguous 256 B
// line; row statistics are wave reductions.  256 threads = 4 rows per workgroup and pass.
#include "coda_align_loss.h"
#include "common.hip.h"

namespace coda {
namespace {

constexpr int kWaves = 4;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

struct AlignParams {
  const float *emb, *gt, *wmask, *text, *logit_scale, *conf, *g;
  const int64_t *labels;
  float *partial, *demb;
  long long ld_l, ld_b, ld_q;
  int nl, b, nq, e, ncls;
};

// Shared forward part for one row: returns (through refs) the row in registers, its norm terms,
// the class logits' log-sum-exp and the label logit.
template <int NV, bool BWD>
__global__ __launch_bounds__(kWaves *kWave) void align_loss_kernel(AlignParams p) {
  const int lane = lane_id(), w = wave_id();
  const long long rows = static_cast<long long>(p.nl) * p.b * p.nq;
  const float t = *p.logit_scale;
  for (long long row = static_cast<long long>(blockIdx.x) * kWaves + w; row < rows;
       row += static_cast<long long>(gridDim.x) * kWaves) {
    const int q = static_cast<int>(row % p.nq);
    const int bi = static_cast<int>((row / p.nq) % p.b);
    const int l = static_cast<int>(row / (static_cast<long long>(p.nq) * p.b));
    const float *er = p.emb + l * p.ld_l + bi * p.ld_b + q * p.ld_q;
    const float *gr = p.gt + (static_cast<size_t>(bi) * p.nq + q) * p.e;
    const float wm = p.wmask[static_cast<size_t>(bi) * p.nq + q];
    const float cf = p.conf[row];
    const int label = static_cast<int>(p.labels[row]);
    const float *tb = p.text + static_cast<size_t>(bi) * p.ncls * p.e;

    float ev[NV], gv[NV];
    float n2 = 0.f, l1 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      ev[k] = er[64 * k + lane];
      gv[k] = gr[64 * k + lane];
      n2 += ev[k] * ev[k];
      l1 += fabsf(ev[k] * wm - gv[k] * wm);
    }
    n2 = wsum(n2);
    const float nrm = sqrtf(n2);
    const float inv = 1.0f / (nrm + 1e-32f);

    // logits_j = t * <e, text_j> * inv ; online log-sum-exp over the classes (wave-uniform)
    float mx = -INFINITY, se = 0.f, ly = 0.f;
    for (int j = 0; j < p.ncls; ++j) {
      const float *tj = tb + static_cast<size_t>(j) * p.e;
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) d += ev[k] * tj[64 * k + lane];
      const float lg = wsum(d) * inv * t;
      if (j == label) ly = lg;
      const float nm = fmaxf(mx, lg);
      se = se * __expf(mx - nm) + __expf(lg - nm);
      mx = nm;
    }
    const float lse = mx + __logf(se);

    if (!BWD) {
      l1 = wsum(l1);
      if (lane == 0) {
        p.partial[row * 2 + 0] = l1;
        p.partial[row * 2 + 1] = (lse - ly) * cf;
      }
    } else {
      const float g1 = p.g[l * 2 + 0], g2 = p.g[l * 2 + 1] * cf;
      // d ehat = t * sum_j g2 * (p_j - [j == label]) * text_j
      float dh[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) dh[k] = 0.f;
      if (g2 != 0.f) {
        for (int j = 0; j < p.ncls; ++j) {
          const float *tj = tb + static_cast<size_t>(j) * p.e;
          float d = 0.f;
#pragma unroll
          for (int k = 0; k < NV; ++k) d += ev[k] * tj[64 * k + lane];
          const float lg = wsum(d) * inv * t;
          const float c = g2 * t * (__expf(lg - lse) - (j == label ? 1.f : 0.f));
#pragma unroll
          for (int k = 0; k < NV; ++k) dh[k] += c * tj[64 * k + lane];
        }
      }
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) s += dh[k] * ev[k];
      s = wsum(s);
      // y = e / (n + eps):  de = dy / (n + eps) - e * <dy, e> / (n * (n + eps)^2)
      const float c2 = nrm > 0.f ? s * inv * inv / nrm : 0.f;
      float *dr = p.demb + row * p.e;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float diff = ev[k] * wm - gv[k] * wm;
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        dr[64 * k + lane] = g1 * sg * wm + dh[k] * inv - ev[k] * c2;
      }
    }
  }
}

// ---- the GEMM route for the stage-2 class counts (232 / 1201 prompts, models/model_3detr.py:321) ------------------------
// With hundreds of classes the row kernel above would spend 2 * ncls * E vector FMAs per row and direction (20 GFLOP per
// direction at 16 384 rows x 1201 x 512); the logits are then ONE dense product ehat (rows x E) . text^T and the gradient
// ONE product dlogits . text, which run on the matrix cores (align_loss.py: gemm.linear / gemm.mm, i.e. the six-product
// bf16x3 kernels of csrc/gemm_x3.hip for >= 8192 rows), and the row-wise pieces are three small kernels:
//   rows_fwd:  l1 partial, n = |e|, ehat = e / (n + 1e-32) written densely in (l, b, q) row order
//   ce:        per row of the logits: lse over the first ncls columns of t * logit, CE partial; backward form: the row is
//              overwritten with g2 * t * (softmax - onehot), zero in the padding columns
//   rows_bwd:  de = g1 * sign(e w - gt w) * w + dh / (n + eps) - e <dh, e> / (n (n + eps)^2)
struct AlignRowsParams {
  const float *emb, *gt, *wmask, *g, *dh;
  float *ehat, *stat, *partial, *demb;  // stat (rows, 2): n, 1 / (n + eps)
  long long ld_l, ld_b, ld_q;
  int nl, b, nq, e;
};

template <int NV, bool BWD>
__global__ __launch_bounds__(kWaves *kWave) void align_rows_kernel(AlignRowsParams p) {
  const int lane = lane_id(), w = wave_id();
  const long long rows = static_cast<long long>(p.nl) * p.b * p.nq;
  for (long long row = static_cast<long long>(blockIdx.x) * kWaves + w; row < rows;
       row += static_cast<long long>(gridDim.x) * kWaves) {
    const int q = static_cast<int>(row % p.nq);
    const int bi = static_cast<int>((row / p.nq) % p.b);
    const int l = static_cast<int>(row / (static_cast<long long>(p.nq) * p.b));
    const float *er = p.emb + l * p.ld_l + bi * p.ld_b + q * p.ld_q;
    const float *gr = p.gt + (static_cast<size_t>(bi) * p.nq + q) * p.e;
    const float wm = p.wmask[static_cast<size_t>(bi) * p.nq + q];
    float ev[NV], gv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      ev[k] = er[64 * k + lane];
      gv[k] = gr[64 * k + lane];
    }
    if (!BWD) {
      float n2 = 0.f, l1 = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        n2 += ev[k] * ev[k];
        l1 += fabsf(ev[k] * wm - gv[k] * wm);
      }
      n2 = wsum(n2);
      l1 = wsum(l1);
      const float nrm = sqrtf(n2), inv = 1.0f / (nrm + 1e-32f);
      float *hr = p.ehat + row * p.e;
#pragma unroll
      for (int k = 0; k < NV; ++k) hr[64 * k + lane] = ev[k] * inv;
      if (lane == 0) {
        p.stat[row * 2 + 0] = nrm;
        p.stat[row * 2 + 1] = inv;
        p.partial[row * 2 + 0] = l1;
      }
    } else {
      const float nrm = p.stat[row * 2 + 0], inv = p.stat[row * 2 + 1];
      const float g1 = p.g[l * 2 + 0];
      const float *hr = p.dh + row * p.e;
      float dh[NV], s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        dh[k] = hr[64 * k + lane];
        s += dh[k] * ev[k];
      }
      s = wsum(s);
      const float c2 = nrm > 0.f ? s * inv * inv / nrm : 0.f;
      float *dr = p.demb + row * p.e;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float diff = ev[k] * wm - gv[k] * wm;
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        dr[64 * k + lane] = g1 * sg * wm + dh[k] * inv - ev[k] * c2;
      }
    }
  }
}

// one wave per logits row (ld floats apart, the first ncls columns are classes)
template <bool BWD>
__global__ __launch_bounds__(kWaves *kWave) void align_ce_kernel(float *__restrict__ logits, long long ld, int ncls, int ncols,
                                                                 const float *__restrict__ logit_scale,
                                                                 const int64_t *__restrict__ labels,
                                                                 const float *__restrict__ conf, const float *__restrict__ g,
                                                                 float *__restrict__ partial, long long rows,
                                                                 long long rows_per_layer) {
  const int lane = lane_id(), w = wave_id();
  const float t = *logit_scale;
  for (long long row = static_cast<long long>(blockIdx.x) * kWaves + w; row < rows;
       row += static_cast<long long>(gridDim.x) * kWaves) {
    float *lr = logits + row * ld;
    const int label = static_cast<int>(labels[row]);
    const float cf = conf[row];
    float mx = -INFINITY;
    for (int j = lane; j < ncls; j += kWave) mx = fmaxf(mx, lr[j] * t);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, kWave));
    float se = 0.f;
    for (int j = lane; j < ncls; j += kWave) se += __expf(lr[j] * t - mx);
    se = wsum(se);
    const float lse = mx + __logf(se);
    if (!BWD) {
      if (lane == 0) partial[row * 2 + 1] = (lse - lr[label] * t) * cf;
    } else {
      const float g2 = g[(row / rows_per_layer) * 2 + 1] * cf;
      for (int j = lane; j < ncols; j += kWave)
        lr[j] = j < ncls ? g2 * t * (__expf(lr[j] * t - lse) - (j == label ? 1.f : 0.f)) : 0.f;
    }
  }
}

bool bad(int nl, int b, int nq, int e, int ncls) {
  return nl < 0 || b < 0 || nq < 0 || e <= 0 || (e % 64) != 0 || e > 1024 || ncls <= 0;
}

template <bool BWD>
int launch(const AlignParams &p, hipStream_t s) {
  const long long rows = static_cast<long long>(p.nl) * p.b * p.nq;
  const long long want = (rows + kWaves - 1) / kWaves;
  const dim3 grid(static_cast<unsigned>(want > 8192 ? 8192 : want));
  clear_sticky_error();
  switch (p.e / 64) {
#define CODA_ALIGN_CASE(N) \
  case N: hipLaunchKernelGGL((align_loss_kernel<N, BWD>), grid, dim3(kWaves * kWave), 0, s, p); break;
    CODA_ALIGN_CASE(1) CODA_ALIGN_CASE(2) CODA_ALIGN_CASE(3) CODA_ALIGN_CASE(4) CODA_ALIGN_CASE(5) CODA_ALIGN_CASE(6)
    CODA_ALIGN_CASE(7) CODA_ALIGN_CASE(8) CODA_ALIGN_CASE(9) CODA_ALIGN_CASE(10) CODA_ALIGN_CASE(11) CODA_ALIGN_CASE(12)
    CODA_ALIGN_CASE(13) CODA_ALIGN_CASE(14) CODA_ALIGN_CASE(15) CODA_ALIGN_CASE(16)
#undef CODA_ALIGN_CASE
    default: return CODA_EINVAL;
  }
  return launch_status();
}

}  // namespace
}  // namespace coda

using namespace coda;

CODA_API int coda_align_loss_fwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q,
                                     const float *gt, const float *wmask, const float *text,
                                     const float *logit_scale, const int64_t *labels, const float *conf, int nl,
                                     int b, int nq, int e, int ncls, float *partial, void *stream) {
  if (bad(nl, b, nq, e, ncls)) return CODA_EINVAL;
  if (static_cast<long long>(nl) * b * nq == 0) return CODA_OK;
  if (!emb || !gt || !wmask || !text || !logit_scale || !labels || !conf || !partial) return CODA_EINVAL;
  AlignParams p{emb, gt, wmask, text, logit_scale, conf, nullptr, labels, partial, nullptr, ld_l, ld_b, ld_q,
                nl, b, nq, e, ncls};
  return launch<false>(p, static_cast<hipStream_t>(stream));
}

CODA_API int coda_align_loss_bwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q,
                                     const float *gt, const float *wmask, const float *text,
                                     const float *logit_scale, const int64_t *labels, const float *conf,
                                     const float *g, int nl, int b, int nq, int e, int ncls, float *demb,
                                     void *stream) {
  if (bad(nl, b, nq, e, ncls)) return CODA_EINVAL;
  if (static_cast<long long>(nl) * b * nq == 0) return CODA_OK;
  if (!emb || !gt || !wmask || !text || !logit_scale || !labels || !conf || !g || !demb) return CODA_EINVAL;
  AlignParams p{emb, gt, wmask, text, logit_scale, conf, g, labels, nullptr, demb, ld_l, ld_b, ld_q,
                nl, b, nq, e, ncls};
  return launch<true>(p, static_cast<hipStream_t>(stream));
}

template <bool BWD>
static int launch_rows(const AlignRowsParams &p, hipStream_t s) {
  const long long rows = static_cast<long long>(p.nl) * p.b * p.nq;
  const long long want = (rows + kWaves - 1) / kWaves;
  const dim3 grid(static_cast<unsigned>(want > 8192 ? 8192 : want));
  clear_sticky_error();
  switch (p.e / 64) {
#define CODA_ALIGN_CASE(N) \
  case N: hipLaunchKernelGGL((align_rows_kernel<N, BWD>), grid, dim3(kWaves * kWave), 0, s, p); break;
    CODA_ALIGN_CASE(1) CODA_ALIGN_CASE(2) CODA_ALIGN_CASE(3) CODA_ALIGN_CASE(4) CODA_ALIGN_CASE(5) CODA_ALIGN_CASE(6)
    CODA_ALIGN_CASE(7) CODA_ALIGN_CASE(8) CODA_ALIGN_CASE(9) CODA_ALIGN_CASE(10) CODA_ALIGN_CASE(11) CODA_ALIGN_CASE(12)
    CODA_ALIGN_CASE(13) CODA_ALIGN_CASE(14) CODA_ALIGN_CASE(15) CODA_ALIGN_CASE(16)
#undef CODA_ALIGN_CASE
    default: return CODA_EINVAL;
  }
  return launch_status();
}

CODA_API int coda_align_rows_fwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q, const float *gt,
                                     const float *wmask, int nl, int b, int nq, int e, float *ehat, float *stat,
                                     float *partial, void *stream) {
  if (bad(nl, b, nq, e, 1)) return CODA_EINVAL;
  if (static_cast<long long>(nl) * b * nq == 0) return CODA_OK;
  if (!emb || !gt || !wmask || !ehat || !stat || !partial) return CODA_EINVAL;
  AlignRowsParams p{emb, gt, wmask, nullptr, nullptr, ehat, stat, partial, nullptr, ld_l, ld_b, ld_q, nl, b, nq, e};
  return launch_rows<false>(p, static_cast<hipStream_t>(stream));
}

CODA_API int coda_align_rows_bwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q, const float *gt,
                                     const float *wmask, const float *stat, const float *dh, const float *g, int nl, int b,
                                     int nq, int e, float *demb, void *stream) {
  if (bad(nl, b, nq, e, 1)) return CODA_EINVAL;
  if (static_cast<long long>(nl) * b * nq == 0) return CODA_OK;
  if (!emb || !gt || !wmask || !stat || !dh || !g || !demb) return CODA_EINVAL;
  AlignRowsParams p{emb, gt, wmask, g, dh, nullptr, const_cast<float *>(stat), nullptr, demb, ld_l, ld_b, ld_q, nl, b, nq, e};
  return launch_rows<true>(p, static_cast<hipStream_t>(stream));
}

CODA_API int coda_align_ce_f32(float *logits, long long ld, int ncls, int ncols, const float *logit_scale,
                               const int64_t *labels, const float *conf, const float *g, float *partial, long long rows,
                               long long rows_per_layer, void *stream) {
  if (rows < 0 || ncls <= 0 || ncols < ncls || ld < ncols || rows_per_layer <= 0) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!logits || !logit_scale || !labels || !conf || (!g && !partial)) return CODA_EINVAL;
  const long long want = (rows + kWaves - 1) / kWaves;
  const dim3 grid(static_cast<unsigned>(want > 16384 ? 16384 : want));
  clear_sticky_error();
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (g) hipLaunchKernelGGL(align_ce_kernel<true>, grid, dim3(kWaves * kWave), 0, s, logits, ld, ncls, ncols, logit_scale,
                            labels, conf, g, partial, rows, rows_per_layer);
  else hipLaunchKernelGGL(align_ce_kernel<false>, grid, dim3(kWaves * kWave), 0, s, logits, ld, ncls, ncols, logit_scale,
                          labels, conf, g, partial, rows, rows_per_layer);
  return launch_status();
}
