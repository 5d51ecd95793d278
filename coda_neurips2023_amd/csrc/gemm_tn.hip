// gemm_tn.hip -- EXPERIMENTAL (off by default, not yet run on hardware): weight-gradient GEMM
//   out (Co x Ci) [+]= dY^T X,   dY (rows x Co), X (rows x Ci),  fp32 MFMA, split over the rows.
//
// Why: the decoder's weight gradients reduce 2048 token rows into 256 x 256 outputs.  The library runs
// them as 64 workgroups with one long K loop (MT32x32x128: 15 us, 18 TFLOP/s, 72 launches per step =
// 1.1 ms); for the 16 384-row encoder shapes the host side splits the rows into a batched GEMM plus a
// sum (two launches).  Here a workgroup owns a 32 x 64 output tile and ONE chunk of the rows: its 4 waves
// interleave over the chunk's row pairs (v_mfma_f32_32x32x2f32: k = 2 rows per instruction, operands read
// straight from global memory -- both are row-contiguous, a wave reads 128 B per half), sum their
// accumulators through LDS and add the tile to `out` with hardware fp32 atomics.  `out` is zeroed first
// unless the caller accumulates; the summation order across row chunks is not fixed (like the fp64
// statistics of the batch-norm kernels), the result differs from a single GEMM in rounding only.
#include "coda_gemm.h"
#include "common.hip.h"

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTnWaves = 4;
constexpr int kTnTile = 32;

// accumulator register r of the lane half `half` holds output row crow(r, half) (32x32 MFMA layout)
__device__ __forceinline__ int crow_tn(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int NT>
__global__ __launch_bounds__(kTnWaves *kWave) void gemm_tn_kernel(const float *__restrict__ dy,
                                                                  const float *__restrict__ x,
                                                                  float *__restrict__ out, int rows, int co, int ci,
                                                                  long long lddy, long long ldx, long long ldout,
                                                                  int rows_per_block) {
  __shared__ float s_part[kTnWaves][NT * 16][kWave];
  const int lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_i = co / kTnTile;
  const int i0 = (static_cast<int>(blockIdx.x) % tiles_i) * kTnTile;
  const int j0 = (static_cast<int>(blockIdx.x) / tiles_i) * kTnTile * NT;
  const int r_begin = static_cast<int>(blockIdx.y) * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // wave w takes the row pairs w, w + 4, w + 8, ... of the chunk; lane half `half` reads row pair[half]
  constexpr int kUnroll = 8;  // row pairs whose loads are in flight together (the short chunks are latency-bound)
  for (int base = r_begin + 2 * w; base < r_end; base += 2 * kTnWaves * kUnroll) {
    float a[kUnroll], b[kUnroll][NT];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int row = base + u * 2 * kTnWaves + half;
      const bool ok = row < r_end;
      a[u] = ok ? dy[static_cast<size_t>(row) * lddy + i0 + l31] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) b[u][t] = ok ? x[static_cast<size_t>(row) * ldx + j0 + t * kTnTile + l31] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
  }

#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_part[w][t * 16 + r][lane] = acc[t][r];
  __syncthreads();
  // wave w finalises accumulator registers 4w .. 4w+3 of every tile: lanes = 32 consecutive columns
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * w + rr;
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < kTnWaves; ++ww) v += s_part[ww][t * 16 + r][lane];
      const int row = i0 + crow_tn(r, half), col = j0 + t * kTnTile + l31;
      atomicAdd(out + static_cast<size_t>(row) * ldout + col, v);
    }
}

template <int NT>
int launch_tn(const float *dy, const float *x, float *out, int rows, int co, int ci, long long lddy, long long ldx,
              long long ldout, hipStream_t s) {
  const int tiles = (co / kTnTile) * (ci / (kTnTile * NT));
  // enough row chunks to put ~2 workgroups on every CU, at least 64 rows (8 row pairs per wave) each
  int splits = max(1, min((rows + 63) / 64, (512 + tiles - 1) / tiles));
  int rows_per_block = ((rows + splits - 1) / splits + 7) & ~7;
  splits = (rows + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((gemm_tn_kernel<NT>), dim3(tiles, splits), dim3(kTnWaves * kWave), 0, s, dy, x, out, rows, co, ci,
                     lddy, ldx, ldout, rows_per_block);
  return launch_status();
}

}  // namespace
}  // namespace coda

CODA_API int coda_gemm_tn_f32(const float *dy, const float *x, float *out, int rows, int co, int ci,
                              long long lddy, long long ldx, long long ldout, int accumulate, void *stream) {
  using namespace coda;
  if (rows < 0 || co <= 0 || ci <= 0 || co % kTnTile != 0 || ci % kTnTile != 0 || lddy < co || ldx < ci || ldout < ci)
    return CODA_EINVAL;
  if (!out || (rows > 0 && (!dy || !x))) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  if (!accumulate) {
    hipError_t e = ldout == ci ? hipMemsetAsync(out, 0, sizeof(float) * static_cast<size_t>(co) * ci, s)
                               : hipMemset2DAsync(out, sizeof(float) * ldout, 0, sizeof(float) * ci, co, s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  if (rows == 0) return CODA_OK;
  return (ci % (2 * kTnTile) == 0) ? launch_tn<2>(dy, x, out, rows, co, ci, lddy, ldx, ldout, s)
                                   : launch_tn<1>(dy, x, out, rows, co, ci, lddy, ldx, ldout, s);
}
