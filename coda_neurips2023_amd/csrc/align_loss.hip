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
