// group_gather.hip -- index gathers and their scatter-add adjoints for gfx950.
//
// Replaces gather_points(_grad) (third_party_pointnet2/pointnet2/_ext_src/src/
// sampling_gpu.cu:11-60) and group_points(_grad) (src/group_points_gpu.cu:11-78).
// The reference launches ONE block per scene and re-reads idx once per channel;
// here the grid covers (index elements x channel chunks x scenes), every thread
// reads its index once (coalesced), keeps CH independent gathers in flight and
// writes CH coalesced output streams.  All four are HBM/L2-bound byte movers.
#include "common.hip.h"

#include <algorithm>
#include <cmath>

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

// ---- deterministic scatter-add (round 5) -----------------------------------------------------------------------------
// The adjoints above sum colliding gradients with hardware float atomics: like the reference's atomicAdd
// (group_points_gpu.cu:46-67, sampling_gpu.cu:37-60) the result depends on the order the atomics happen in, i.e. it is
// not reproducible from run to run.  Integer addition is associative: every addend is converted to 64-bit fixed point
// at a scale taken from the largest magnitude of its (scene, channel) row (so that the worst-case sum of all e_count addends still fits),
// accumulated with 64-bit integer atomics -- ANY order gives the same bits -- and converted back once.  The fixed-point
// grid has 62 - ceil(log2(e_count + 1)) bits below the largest magnitude (47 at 32 768 entries per scene, never fewer
// than 31): finer than the float32 roundings of a float accumulation.  Three passes (largest magnitude, scatter,
// convert) and 8 bytes of workspace per output element.  A non-finite gradient makes its (scene, channel) row NaN
// (the float atomics would have poisoned only the targets it reaches; a training step is lost either way).
// One scale per (scene, channel) ROW of the output (round 6; until then one for the whole tensor: a channel or a scene
// whose gradients sat more than ~2^-(62 - count_bits) below the global maximum rounded to zero, and one non-finite value
// anywhere turned the entire result into NaN -- ADVICE r5).  The header holds the float bits of max |x| per row
// (non-negative floats order like their bit patterns), padded to a 256-byte multiple in front of the accumulators.
__host__ __device__ inline size_t det_header_bytes(size_t rows) { return (sizeof(unsigned int) * rows + 255) & ~static_cast<size_t>(255); }

// grid (chunks of the row, rows)
__global__ __launch_bounds__(kThreads) void det_absmax_kernel(const float *__restrict__ x, int e_count,
                                                              unsigned int *__restrict__ row_max) {
  const float *__restrict__ r = x + static_cast<size_t>(blockIdx.y) * e_count;
  unsigned int m = 0u;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < e_count; i += gridDim.x * kThreads)
    m = max(m, __float_as_uint(r[i]) & 0x7fffffffu);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, static_cast<unsigned int>(__shfl_xor(static_cast<int>(m), off)));
  if ((threadIdx.x & (kWave - 1)) == 0 && m != 0u) atomicMax(&row_max[blockIdx.y], m);
}
// 2^k with  max |x| * 2^k < 2^(62 - count_bits):  k = 62 - count_bits - (exponent of max + 1)
__device__ __forceinline__ double det_scale(unsigned int absmax_bits, int count_bits) {
  if (absmax_bits == 0u) return 1.0;
  const int e = static_cast<int>(absmax_bits >> 23) - 127;  // max < 2^(e + 1)  (denormals: e = -127, still an upper bound)
  return ldexp(1.0, 62 - count_bits - (e + 1));
}
__global__ __launch_bounds__(kThreads) void det_scatter_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                                               const unsigned int *__restrict__ row_max,
                                                               unsigned long long *__restrict__ acc, int c, int n, int e_count,
                                                               int count_bits) {
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= e_count) return;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * kCh;
  const int a = idx[static_cast<size_t>(bi) * e_count + e];
  const float *__restrict__ src = grad_out + (static_cast<size_t>(bi) * c + c0) * e_count + e;
  unsigned long long *__restrict__ dst = acc + (static_cast<size_t>(bi) * c + c0) * n + a;
#pragma unroll
  for (int l = 0; l < kCh; ++l)
    if (c0 + l < c) {
      const unsigned int mb = row_max[static_cast<size_t>(bi) * c + c0 + l];
      if (mb >= 0x7f800000u) continue;  // a non-finite value in this row: the convert pass writes NaN into the row
      const long long q = __double2ll_rn(static_cast<double>(src[static_cast<size_t>(l) * e_count]) * det_scale(mb, count_bits));
      if (q != 0) atomicAdd(dst + static_cast<size_t>(l) * n, static_cast<unsigned long long>(q));  // two's complement
    }
}
__global__ __launch_bounds__(kThreads) void det_convert_kernel(const unsigned long long *__restrict__ acc,
                                                               const unsigned int *__restrict__ row_max, float *__restrict__ out,
                                                               size_t total, int n, int count_bits) {
  const size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= total) return;
  const unsigned int mb = row_max[i / n];
  if (mb >= 0x7f800000u) {
    out[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  out[i] = static_cast<float>(static_cast<double>(static_cast<long long>(acc[i])) / det_scale(mb, count_bits));
}

size_t det_workspace_bytes(int b, int c, int n) {
  return det_header_bytes(static_cast<size_t>(b) * c) + sizeof(unsigned long long) * static_cast<size_t>(b) * c * n;
}

int scatter_add_rows_det(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c, int n, long long e_count,
                         void *workspace, size_t workspace_bytes, hipStream_t s) {
  if (e_count > 0x7fffffffLL) return CODA_EINVAL;
  const size_t out_total = static_cast<size_t>(b) * c * n;
  if (e_count == 0) {
    const hipError_t e0 = hipMemsetAsync(grad_points, 0, sizeof(float) * out_total, s);
    return e0 == hipSuccess ? CODA_OK : static_cast<int>(e0);
  }
  if (!workspace || workspace_bytes < det_workspace_bytes(b, c, n) || (reinterpret_cast<uintptr_t>(workspace) & 7) != 0)
    return CODA_ENOSPC;
  const size_t rows = static_cast<size_t>(b) * c;
  unsigned int *row_max = static_cast<unsigned int *>(workspace);
  unsigned long long *acc = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + det_header_bytes(rows));
  const hipError_t e = hipMemsetAsync(workspace, 0, det_workspace_bytes(b, c, n), s);
  if (e != hipSuccess) return static_cast<int>(e);
  int count_bits = 1;
  while ((1LL << count_bits) <= e_count) ++count_bits;  // ceil(log2(e_count + 1))
  clear_sticky_error();
  if (rows > 65535) return CODA_EINVAL;  // (grid.y; far beyond any (scene, channel) count of the path)
  const unsigned int chunks = static_cast<unsigned int>(std::min<long long>((e_count + kThreads - 1) / kThreads, 64));
  hipLaunchKernelGGL(det_absmax_kernel, dim3(chunks, static_cast<unsigned int>(rows)), dim3(kThreads), 0, s, grad_out,
                     static_cast<int>(e_count), row_max);
  dim3 grid(ceil_div(static_cast<int>(e_count), kThreads), ceil_div(c, kCh), b);
  hipLaunchKernelGGL(det_scatter_kernel, grid, dim3(kThreads), 0, s, grad_out, idx, row_max, acc, c, n, static_cast<int>(e_count),
                     count_bits);
  hipLaunchKernelGGL(det_convert_kernel, dim3(static_cast<unsigned int>((out_total + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     acc, row_max, grad_points, out_total, n, count_bits);
  return launch_status();
}

}  // namespace
}  // namespace coda

CODA_API size_t coda_scatter_add_det_workspace_bytes(int b, int c, int n) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  return coda::det_workspace_bytes(b, c, n);
}

CODA_API int coda_gather_points_grad_det_f32(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c, int n,
                                             int m, void *workspace, size_t workspace_bytes, void *stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || n == 0) return CODA_OK;
  if (!grad_points || (m > 0 && (!grad_out || !idx))) return CODA_EINVAL;
  return coda::scatter_add_rows_det(grad_out, idx, grad_points, b, c, n, m, workspace, workspace_bytes,
                                    static_cast<hipStream_t>(stream));
}

CODA_API int coda_group_points_grad_det_f32(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c, int n,
                                            int npoints, int nsample, void *workspace, size_t workspace_bytes, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || n == 0) return CODA_OK;
  const long long e_count = static_cast<long long>(npoints) * nsample;
  if (!grad_points || (e_count > 0 && (!grad_out || !idx))) return CODA_EINVAL;
  return coda::scatter_add_rows_det(grad_out, idx, grad_points, b, c, n, e_count, workspace, workspace_bytes,
                                    static_cast<hipStream_t>(stream));
}

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
