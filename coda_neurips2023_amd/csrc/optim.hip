// optim.hip -- gradient clipping + AdamW over all parameters of a group in three launches (include/coda_optim.h):
// what engine.py:161-164 does per step with torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW.step.
//
// A parameter list is described once by a chunk map (chunk -> tensor, offset): every 256-thread block owns kChunk
// consecutive elements of one tensor, so 252 tensors from 2 to 524 288 elements fill the chip with one launch.
// HBM-bound: the update reads p, g, m, v and writes p, m, v once.
#include "coda_optim.h"
#include "common.hip.h"

#include <math.h>

namespace coda {
namespace {

constexpr int kChunk = 2048;  // elements per block: 8 per thread

__device__ __forceinline__ bool aligned16(const void *a, const void *b, const void *c, const void *d) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

// A block walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... and issues ONE double atomic at the end: with one
// atomic per chunk the 4 000 - 12 000 same-address atomics serialised at the L2 and took 10x the read time.
__global__ __launch_bounds__(256) void sumsq_kernel(const CodaOptTensor *__restrict__ tab, const int2 *__restrict__ chunks,
                                                    int nchunks, double *__restrict__ out) {
  float acc = 0.0f;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int2 ch = chunks[c];
    const CodaOptTensor t = tab[ch.x];
    if (!t.g) continue;
    const long long begin = static_cast<long long>(ch.y) * kChunk;
    const long long end = begin + kChunk < t.n ? begin + kChunk : t.n;
    if ((reinterpret_cast<uintptr_t>(t.g) & 15) == 0) {
      for (long long i = begin + 4 * threadIdx.x; i < end; i += 1024) {
        if (i + 4 <= end) {
          const float4 v = *reinterpret_cast<const float4 *>(t.g + i);
          acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
          for (long long j = i; j < end; ++j) acc += t.g[j] * t.g[j];
        }
      }
    } else {
      for (long long i = begin + threadIdx.x; i < end; i += 256) acc += t.g[i] * t.g[i];
    }
  }
  __shared__ float s_part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) s_part[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, static_cast<double>(s_part[0]) + s_part[1] + s_part[2] + s_part[3]);
}

// torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1; every gradient is multiplied by it
__global__ __launch_bounds__(256) void scale_kernel(const CodaOptTensor *__restrict__ tab, const int2 *__restrict__ chunks,
                                                    const double *__restrict__ sumsq, float max_norm,
                                                    float *__restrict__ total_norm) {
  const float norm = static_cast<float>(sqrt(*sumsq));
  if (blockIdx.x == 0 && threadIdx.x == 0 && total_norm) *total_norm = norm;
  float coef = max_norm / (norm + 1e-6f);
  coef = coef > 1.0f ? 1.0f : coef;
  const int2 ch = chunks[blockIdx.x];
  const CodaOptTensor t = tab[ch.x];
  if (!t.g) return;
  const long long begin = static_cast<long long>(ch.y) * kChunk;
  const long long end = begin + kChunk < t.n ? begin + kChunk : t.n;
  for (long long i = begin + threadIdx.x; i < end; i += 256) t.g[i] = t.g[i] * coef;
}

// Gradient packing for the data-parallel all-reduce: dst (tab.p) = src (tab.g) * scale, zeros where a tensor has no
// gradient this step.  One launch for all tensors (torch's DistributedDataParallel: one copy kernel per tensor).
__global__ __launch_bounds__(256) void pack_kernel(const CodaOptTensor *__restrict__ tab, const int2 *__restrict__ chunks,
                                                   float scale) {
  const int2 ch = chunks[blockIdx.x];
  const CodaOptTensor t = tab[ch.x];
  const long long begin = static_cast<long long>(ch.y) * kChunk;
  const long long end = begin + kChunk < t.n ? begin + kChunk : t.n;
  if (t.g && aligned16(t.p, t.g, nullptr, nullptr)) {
    for (long long i = begin + 4 * threadIdx.x; i < end; i += 1024) {
      if (i + 4 <= end) {
        float4 v = *reinterpret_cast<const float4 *>(t.g + i);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4 *>(t.p + i) = v;
      } else {
        for (long long j = i; j < end; ++j) t.p[j] = t.g[j] * scale;
      }
    }
  } else {
    for (long long i = begin + threadIdx.x; i < end; i += 256) t.p[i] = t.g ? t.g[i] * scale : 0.0f;
  }
}

struct AdamArgs { float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt; };
struct AdamHyper { float lr, beta1, beta2, eps, weight_decay; };

__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, const AdamArgs &a) {
  if (a.weight_decay != 0.0f) p -= a.lr * a.weight_decay * p;
  m = m + (1.0f - a.beta1) * (g - m);
  v = a.beta2 * v + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bias_correction2_sqrt + a.eps;
  p -= (a.lr / a.bias_correction1) * m / denom;
}

__global__ __launch_bounds__(256) void adamw_kernel(const CodaOptTensor *__restrict__ tab, const int2 *__restrict__ chunks,
                                                    AdamHyper h) {
  const int2 ch = chunks[blockIdx.x];
  const CodaOptTensor t = tab[ch.x];
  if (!t.g) return;  // parameter without a gradient this step: untouched, like torch's optimizers
  __shared__ float s_bc[2];
  if (threadIdx.x == 0) {  // bias corrections of THIS tensor's step count, in double like the host-side formula
    s_bc[0] = static_cast<float>(1.0 - pow(static_cast<double>(h.beta1), t.step));
    s_bc[1] = static_cast<float>(sqrt(1.0 - pow(static_cast<double>(h.beta2), t.step)));
  }
  __syncthreads();
  AdamArgs a;
  a.lr = h.lr; a.beta1 = h.beta1; a.beta2 = h.beta2; a.eps = h.eps; a.weight_decay = h.weight_decay;
  a.bias_correction1 = s_bc[0];
  a.bias_correction2_sqrt = s_bc[1];
  const long long begin = static_cast<long long>(ch.y) * kChunk;
  const long long end = begin + kChunk < t.n ? begin + kChunk : t.n;
  if (aligned16(t.p, t.g, t.m, t.v)) {
    for (long long i = begin + 4 * threadIdx.x; i < end; i += 1024) {
      if (i + 4 <= end) {
        float4 p = *reinterpret_cast<float4 *>(t.p + i), m = *reinterpret_cast<float4 *>(t.m + i),
               v = *reinterpret_cast<float4 *>(t.v + i);
        const float4 g = *reinterpret_cast<const float4 *>(t.g + i);
        adam_update(p.x, g.x, m.x, v.x, a); adam_update(p.y, g.y, m.y, v.y, a);
        adam_update(p.z, g.z, m.z, v.z, a); adam_update(p.w, g.w, m.w, v.w, a);
        *reinterpret_cast<float4 *>(t.p + i) = p;
        *reinterpret_cast<float4 *>(t.m + i) = m;
        *reinterpret_cast<float4 *>(t.v + i) = v;
      } else {
        for (long long j = i; j < end; ++j) adam_update(t.p[j], t.g[j], t.m[j], t.v[j], a);
      }
    }
  } else {
    for (long long i = begin + threadIdx.x; i < end; i += 256) adam_update(t.p[i], t.g[i], t.m[i], t.v[i], a);
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_opt_chunk_elems(void) { return coda::kChunk; }

CODA_API int coda_opt_grad_sumsq_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, double *sumsq,
                                     void *stream) {
  using namespace coda;
  if (nchunks < 0 || !sumsq) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const hipError_t e = hipMemsetAsync(sumsq, 0, sizeof(double), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (nchunks == 0) return CODA_OK;
  if (!table || !chunks) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(sumsq_kernel, dim3(nchunks < 512 ? nchunks : 512), dim3(256), 0, s, table,
                     reinterpret_cast<const int2 *>(chunks), nchunks, sumsq);
  return launch_status();
}

CODA_API int coda_opt_grad_scale_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, const double *sumsq,
                                     float max_norm, float *total_norm, void *stream) {
  using namespace coda;
  if (nchunks < 0 || !sumsq || !(max_norm >= 0.0f)) return CODA_EINVAL;
  if (nchunks == 0) return CODA_OK;
  if (!table || !chunks) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(scale_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream), table,
                     reinterpret_cast<const int2 *>(chunks), sumsq, max_norm, total_norm);
  return launch_status();
}

CODA_API int coda_opt_adamw_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, float lr, float beta1,
                                float beta2, float eps, float weight_decay, void *stream) {
  using namespace coda;
  if (nchunks < 0 || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f)) return CODA_EINVAL;
  if (nchunks == 0) return CODA_OK;
  if (!table || !chunks) return CODA_EINVAL;
  AdamHyper a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  clear_sticky_error();
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream), table,
                     reinterpret_cast<const int2 *>(chunks), a);
  return launch_status();
}

CODA_API int coda_opt_pack_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, float scale, void *stream) {
  using namespace coda;
  if (nchunks < 0) return CODA_EINVAL;
  if (nchunks == 0) return CODA_OK;
  if (!table || !chunks) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(pack_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream), table,
                     reinterpret_cast<const int2 *>(chunks), scale);
  return launch_status();
}
