// interpolate.hip -- three_nn / three_interpolate(_grad) for gfx950.
//
// Replaces third_party_pointnet2/pointnet2/_ext_src/src/interpolate_gpu.cu:
// three_nn_kernel :12-71, three_interpolate_kernel :75-115,
// three_interpolate_grad_kernel :119-158 (each one block per scene there).
//
// three_nn: lane = one `unknown` point, the `known` cloud is streamed through
// LDS in tiles and read back as wave-uniform broadcasts, ascending k, so the
// strict-`<` insertion (:37-52) reproduces the reference's tie order (lowest k
// first) without any merge step.  The reference keeps its three running bests
// in double, initialised 1e40 (:30); they only ever hold float values or the
// initial 1e40, so float bests initialised +inf compare identically and the
// final `(float)1e40` store equals +inf.
#include "common.hip.h"

namespace coda {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;  // known points per LDS tile (12 KiB)

template <int DM>
__global__ __launch_bounds__(kThreads) void three_nn_kernel(const float *__restrict__ unknown,
                                                            const float *__restrict__ known,
                                                            float *__restrict__ dist2,
                                                            int32_t *__restrict__ idx, int n,
                                                            int m) {
  __shared__ float s_known[kTile * 3];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const bool active = j < n;
  const float *__restrict__ U = unknown + (static_cast<size_t>(bi) * n + (active ? j : 0)) * 3;
  const float *__restrict__ K = known + static_cast<size_t>(bi) * m * 3;
  const float ux = U[0], uy = U[1], uz = U[2];

  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += kTile) {
    const int cnt = min(kTile, m - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += kThreads) s_known[t] = K[static_cast<size_t>(k0) * 3 + t];
    __syncthreads();
    for (int kk = 0; kk < cnt; ++kk) {
      const float d = sqdist3<DM>(__fsub_rn(ux, s_known[kk * 3 + 0]), __fsub_rn(uy, s_known[kk * 3 + 1]),
                              __fsub_rn(uz, s_known[kk * 3 + 2]));
      const int k = k0 + kk;
      if (d < best1) {
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d; besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d; besti2 = k;
      } else if (d < best3) {
        best3 = d; besti3 = k;
      }
    }
  }
  if (active) {
    float *D = dist2 + (static_cast<size_t>(bi) * n + j) * 3;
    int32_t *I = idx + (static_cast<size_t>(bi) * n + j) * 3;
    D[0] = best1; D[1] = best2; D[2] = best3;
    I[0] = besti1; I[1] = besti2; I[2] = besti3;
  }
}

// out[b,c,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3   (:101-102), rounded per the distance mode (dot3)
template <int DM>
__global__ __launch_bounds__(kThreads) void three_interpolate_kernel(
    const float *__restrict__ points, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out, int c, int m, int n) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= n) return;
  const int bi = blockIdx.z;
  const int32_t *I = idx + (static_cast<size_t>(bi) * n + j) * 3;
  const float *W = weight + (static_cast<size_t>(bi) * n + j) * 3;
  const int i1 = I[0], i2 = I[1], i3 = I[2];
  const float w1 = W[0], w2 = W[1], w3 = W[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float *P = points + (static_cast<size_t>(bi) * c + l) * m;
    out[(static_cast<size_t>(bi) * c + l) * n + j] =
        dot3<DM>(P[i1], w1, P[i2], w2, P[i3], w3);
  }
}

__global__ __launch_bounds__(kThreads) void three_interpolate_grad_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points, int c, int n, int m) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= n) return;
  const int bi = blockIdx.z;
  const int32_t *I = idx + (static_cast<size_t>(bi) * n + j) * 3;
  const float *W = weight + (static_cast<size_t>(bi) * n + j) * 3;
  const int i1 = I[0], i2 = I[1], i3 = I[2];
  const float w1 = W[0], w2 = W[1], w3 = W[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float g = grad_out[(static_cast<size_t>(bi) * c + l) * n + j];
    float *G = grad_points + (static_cast<size_t>(bi) * c + l) * m;
    unsafeAtomicAdd(G + i1, __fmul_rn(g, w1));  // :142-144
    unsafeAtomicAdd(G + i2, __fmul_rn(g, w2));
    unsafeAtomicAdd(G + i3, __fmul_rn(g, w3));
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_three_nn_f32(const float *unknown, const float *known, float *dist2, int32_t *idx,
                               int b, int n, int m, void *stream) {
  using namespace coda;
  if (b < 0 || n < 0 || m < 0) return CODA_EINVAL;
  if (b == 0 || n == 0) return CODA_OK;
  if (!unknown || !dist2 || !idx || (m > 0 && !known)) return CODA_EINVAL;
  clear_sticky_error();
  CODA_DISPATCH_DM(distance_mode(),
                   hipLaunchKernelGGL(three_nn_kernel<DM>, dim3(ceil_div(n, kThreads), b), dim3(kThreads), 0,
                                      static_cast<hipStream_t>(stream), unknown, known, dist2, idx, n, m));
  return launch_status();
}

CODA_API int coda_three_interpolate_f32(const float *points, const int32_t *idx,
                                        const float *weight, float *out, int b, int c, int m,
                                        int n, void *stream) {
  using namespace coda;
  if (b < 0 || c < 0 || n < 0 || m < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || n == 0) return CODA_OK;
  if (!points || !idx || !weight || !out || m == 0) return CODA_EINVAL;
  dim3 grid(ceil_div(n, kThreads), c < 64 ? c : 64, b);
  clear_sticky_error();
  CODA_DISPATCH_DM(distance_mode(),
                   hipLaunchKernelGGL(three_interpolate_kernel<DM>, grid, dim3(kThreads), 0,
                                      static_cast<hipStream_t>(stream), points, idx, weight, out, c, m, n));
  return launch_status();
}

CODA_API int coda_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx,
                                             const float *weight, float *grad_points, int b, int c,
                                             int n, int m, void *stream) {
  using namespace coda;
  if (b < 0 || c < 0 || n < 0 || m < 0) return CODA_EINVAL;
  if (b == 0 || c == 0 || m == 0) return CODA_OK;
  if (!grad_points || (n > 0 && (!grad_out || !idx || !weight))) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(b) * c * m, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return CODA_OK;
  dim3 grid(ceil_div(n, kThreads), c < 64 ? c : 64, b);
  clear_sticky_error();
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(kThreads), 0, s, grad_out, idx,
                     weight, grad_points, c, n, m);
  return launch_status();
}

CODA_API int coda_three_nn_opt_f32(const float *unknown, const float *known, float *dist2, int32_t *idx, int b, int n,
                                   int m, int distance_mode, void *stream) {
  if (distance_mode < -1 || distance_mode >= coda::kDistanceModes) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.distance_mode = distance_mode;
  coda::ScopedCallOptions scope(o);
  return coda_three_nn_f32(unknown, known, dist2, idx, b, n, m, stream);
}

CODA_API int coda_three_interpolate_opt_f32(const float *points, const int32_t *idx, const float *weight, float *out,
                                            int b, int c, int m, int n, int distance_mode, void *stream) {
  if (distance_mode < -1 || distance_mode >= coda::kDistanceModes) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.distance_mode = distance_mode;
  coda::ScopedCallOptions scope(o);
  return coda_three_interpolate_f32(points, idx, weight, out, b, c, m, n, stream);
}
