// pos_embed.hip -- the decoder's Fourier coordinate embedding in one launch (models/position_embedding.py:97-130:
// normalise to the scene's box, times 2*pi, project with the fixed Gaussian matrix, [sin | cos]).  The reference
// spends eight small launches on it (sub, sub, div, mul, mm with K = 3, sin, cos, cat), twice per step.
#include "coda_token_ops.h"
#include "common.hip.h"

namespace coda {
namespace {

// one thread per (point, channel pair c): out[point][c] = sin(phase), out[point][half + c] = cos(phase)
__global__ __launch_bounds__(256) void fourier_pos_embed_kernel(const float *__restrict__ xyz, const float *__restrict__ lo,
                                                                const float *__restrict__ hi, const float *__restrict__ gauss,
                                                                int ldg, float *__restrict__ out, long long points, int n,
                                                                int half) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= points * half) return;
  const long long pt = i / half;
  const int c = static_cast<int>(i - pt * half);
  const int b = static_cast<int>(pt / n);
  float u[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float v = xyz[pt * 3 + a];
    // (p - lo) / (hi - lo), then * 2*pi: the reference's operations, one rounding each (-ffp-contract=off)
    if (lo) v = __fdiv_rn(__fsub_rn(v, lo[b * 3 + a]), __fsub_rn(hi[b * 3 + a], lo[b * 3 + a]));
    u[a] = __fmul_rn(v, 6.283185307179586f);
  }
  const float phase = __fadd_rn(__fadd_rn(__fmul_rn(u[0], gauss[c]), __fmul_rn(u[1], gauss[ldg + c])),
                                __fmul_rn(u[2], gauss[2 * ldg + c]));
  float *row = out + pt * 2 * half;
  row[c] = sinf(phase);
  row[half + c] = cosf(phase);
}

}  // namespace
}  // namespace coda

CODA_API int coda_fourier_pos_embed_f32(const float *xyz, const float *range_lo, const float *range_hi,
                                        const float *gauss, int ld_gauss, float *out, int b, int n, int half,
                                        void *stream) {
  using namespace coda;
  if (b < 0 || n < 0 || half <= 0 || ld_gauss < half || ((range_lo == nullptr) != (range_hi == nullptr))) return CODA_EINVAL;
  const long long points = static_cast<long long>(b) * n;
  if (points == 0) return CODA_OK;
  if (!xyz || !gauss || !out) return CODA_EINVAL;
  const long long work = points * half;
  clear_sticky_error();
  hipLaunchKernelGGL(fourier_pos_embed_kernel, dim3(static_cast<unsigned>((work + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), xyz, range_lo, range_hi, gauss, ld_gauss, out, points, n, half);
  return launch_status();
}
