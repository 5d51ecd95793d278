// group_gather.hip -- index gathers and their scatter-add adjoints for gfx950.
//
// Replaces gather_points(_grad) (third_party_pointnet2/pointnet2/_ext_src/src/
// sampling_gpu.cu:11-60) and group_points(_grad) (src/group_points_gpu.cu:11-78).
// The reference launches ONE block per scene and re-reads idx once per channel;
// here the grid covers (index elements x channel chunks x scenes), every thread
// reads its index once (coalesced), keeps CH independent gathers in flight and
// writes CH coalesced output streams.  All four are HBM/L2-bound byte movers.
#include "common.hip.h"

namespace coda {
namespace {

constexpr int kThreads = 256;
constexpr int kCh = 8;  // channels per thread

// out[b,c,e] = points[b,c,idx[b,e]] for e in [0,E): E = m (gather) or M*S (group).
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float *__restrict__ points,
                                                               const int32_t *__restrict__ idx,
                                                               float *__restrict__ out, int c,
                                                               int n, int e_count) {
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= e_count) return;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * kCh;
  const int a = idx[static_cast<size_t>(bi) * e_count + e];
  const float *__restrict__ src = points + (static_cast<size_t>(bi) * c + c0) * n + a;
  float *__restrict__ dst = out + (static_cast<size_t>(bi) * c + c0) * e_count + e;
  float v[kCh];
#pragma unroll
  for (int l = 0; l < kCh; ++l)
    if (c0 + l < c) v[l] = src[static_cast<size_t>(l) * n];
#pragma unroll
  for (int l = 0; l < kCh; ++l)
    if (c0 + l < c) dst[static_cast<size_t>(l) * e_count] = v[l];
}

// grad_points[b,c,idx[b,e]] += grad_out[b,c,e]
__global__ __launch_bounds__(kThreads) void scatter_add_rows_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    float *__restrict__ grad_points, int c, int n, int e_count) {
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= e_count) return;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * kCh;
  const int a = idx[static_cast<size_t>(bi) * e_count + e];
  const float *__restrict__ src = grad_out + (static_cast<size_t>(bi) * c + c0) * e_count + e;
  float *__restrict__ dst = grad_points + (static_cast<size_t>(bi) * c + c0) * n + a;
#pragma unroll
  for (int l = 0; l < kCh; ++l)
    if (c0 + l < c) unsafeAtomicAdd(dst + static_cast<size_t>(l) * n, src[static_cast<size_t>(l) * e_count]);
}

int gather_rows(const float *points, const int32_t *idx, float *out, int b, int c, int n,
                long long e_count, hipStream_t s) {
  if (e_count > 0x7fffffffLL) return CODA_EINVAL;
  dim3 grid(ceil_div(static_cast<int>(e_count), kThreads), ceil_div(c, kCh), b);
  clear_sticky_error();
  hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(kThreads), 0, s, points, idx, out, c, n,
                     static_cast<int>(e_count));
  return launch_status();
}

int scatter_add_rows(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c,
                     int n, long long e_count, hipStream_t s) {
  if (e_count > 0x7fffffffLL) return CODA_EINVAL;
  hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(b) * c * n, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (e_count == 0) return CODA_OK;
  dim3 grid(ceil_div(static_cast<int>(e_count), kThreads), ceil_div(c, kCh), b);
  clear_sticky_error();
  hipLaunchKernelGGL(scatter_add_rows_kernel, grid, dim3(kThreads), 0, s, grad_out, idx,
                     grad_points, c, n, static_cast<int>(e_count));
  return launch_status();
}

}  // namespace
}  // namespace coda

CODA_API int coda_gather_points_f32(const float *points, const int32_t *idx, float *out, int b,
                                    int c, int n, int m, void *stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || m == 0) return CODA_OK;
  if (!points || !idx || !out || n == 0) return CODA_EINVAL;
  return coda::gather_rows(points, idx, out, b, c, n, m, static_cast<hipStream_t>(stream));
}

CODA_API int coda_gather_points_grad_f32(const float *grad_out, const int32_t *idx,
                                         float *grad_points, int b, int c, int n, int m,
                                         void *stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || n == 0) return CODA_OK;
  if (!grad_points || (m > 0 && (!grad_out || !idx))) return CODA_EINVAL;
  return coda::scatter_add_rows(grad_out, idx, grad_points, b, c, n, m,
                                static_cast<hipStream_t>(stream));
}

CODA_API int coda_group_points_f32(const float *points, const int32_t *idx, float *out, int b,
                                   int c, int n, int npoints, int nsample, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || npoints == 0 || nsample == 0) return CODA_OK;
  if (!points || !idx || !out || n == 0) return CODA_EINVAL;
  return coda::gather_rows(points, idx, out, b, c, n,
                           static_cast<long long>(npoints) * nsample,
                           static_cast<hipStream_t>(stream));
}

CODA_API int coda_group_points_grad_f32(const float *grad_out, const int32_t *idx,
                                        float *grad_points, int b, int c, int n, int npoints,
                                        int nsample, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || n == 0) return CODA_OK;
  const long long e_count = static_cast<long long>(npoints) * nsample;
  if (!grad_points || (e_count > 0 && (!grad_out || !idx))) return CODA_EINVAL;
  return coda::scatter_add_rows(grad_out, idx, grad_points, b, c, n, e_count,
                                static_cast<hipStream_t>(stream));
}
