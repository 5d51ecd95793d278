// gemm_grouped_tn.hip -- many weight-gradient products in ONE launch:  C_p (m_p x n_p) = dy_p^T x_p,
// dy_p (rows_p x m_p), x_p (rows_p x n_p), p = 0 .. count-1.
//
// Why: the decoder stack is one autograd node (fused_blocks._DecoderStack); its backward produced 72 weight
// gradients of 256 x 256 over 2048 token rows one by one, each a 14.5 us library launch at 18 TFLOP/s (a 256 x
// 256 output is 16 tiles: the chip is 94 % idle) -- 1.05 ms of a 19.6 ms step.  Nothing downstream in the backward
// reads a weight gradient, so the node defers them and issues them together at its end: 72 x 16 = 1152 tiles
// fill the 256 CUs.
//
// Shape: one workgroup (4 waves) per 64 x 64 output tile; the token rows are split over the four waves
// (split-K), every wave accumulates the whole tile in 4 x 16 registers with v_mfma_f32_32x32x2_f32 and the
// four partial tiles are summed through LDS in a fixed order (deterministic, no atomics, no zero-fill).  Both
// operands are used exactly as they lie in memory: A[i][k-slot] = dy[k][i0 + i] and B[k-slot][j] = x[k][j0 + j]
// are 128 contiguous bytes per k-slot, so lanes load their operand values straight from global memory (L2) -- no
// LDS staging, no transposition.  Tiles of one problem are dealt to ONE XCD (workgroup g runs on XCD g % 8), so
// a problem's operands are fetched from HBM once and re-read by its other tiles from that XCD's L2.
#include "coda_gemm.h"
#include "common.hip.h"

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kGT = 64;        // tile edge
constexpr int kGThreads = 256;
constexpr int kUnroll = 8;     // k-steps (of 2 rows) in flight per wave
constexpr int kXcds = 8;

struct Problem {
  const float *dy, *x;
  float *out;
  int rows, lddy, ldx, ldout;
  unsigned short tiles_m, tiles_n;  // 64-wide tiles
  int pad_;
};
constexpr int kMaxProblems = 64;  // 48 B each: 3 KB of kernel arguments
struct Batch {
  Problem p[kMaxProblems];
  int count;
  int slots_per_problem;  // tiles of the largest problem in the batch
};

__device__ __forceinline__ int crow_g(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__global__ __launch_bounds__(kGThreads, 2) void grouped_tn_kernel(const Batch batch) {
  __shared__ float s_part[3][kGT * kGT];  // partial tiles of waves 1..3 (wave 0 keeps its own in registers)
  // XCD-aware dealing: workgroup g -> XCD g % 8; a problem's tiles share an XCD
  const int g = blockIdx.x, xcd = g % kXcds, slot = g / kXcds;
  const int pidx = (slot / batch.slots_per_problem) * kXcds + xcd, tile = slot % batch.slots_per_problem;
  if (pidx >= batch.count) return;
  const Problem &pr = batch.p[pidx];
  if (tile >= pr.tiles_m * pr.tiles_n) return;
  const int i0 = (tile / pr.tiles_n) * kGT, j0 = (tile % pr.tiles_n) * kGT;
  const int lane = lane_id(), w = wave_id(), half = lane >> 5, l31 = lane & 31;
  const int per_wave = pr.rows / 4;  // rows is a multiple of 8 (host check): an even count per wave
  const float *pa = pr.dy + static_cast<size_t>(w * per_wave + half) * pr.lddy + i0 + l31;
  const float *pb = pr.x + static_cast<size_t>(w * per_wave + half) * pr.ldx + j0 + l31;
  const size_t sa = 2 * static_cast<size_t>(pr.lddy), sb = 2 * static_cast<size_t>(pr.ldx);
  f32x16 acc[2][2] = {};
  const int steps = per_wave / 2;
  int s = 0;
  for (; s + kUnroll <= steps; s += kUnroll) {
    float a0[kUnroll], a1[kUnroll], b0[kUnroll], b1[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      a0[u] = pa[u * sa];
      a1[u] = pa[u * sa + 32];
      b0[u] = pb[u * sb];
      b1[u] = pb[u * sb + 32];
    }
    pa += kUnroll * sa;
    pb += kUnroll * sb;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
    }
  }
  for (; s < steps; ++s) {
    const float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
    pa += sa;
    pb += sb;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
  // combine: waves 1..3 park their tiles in LDS, wave 0 adds them in wave order and stores
  if (w > 0) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s_part[w - 1][(ti * 32 + crow_g(r, half)) * kGT + tj * 32 + l31] = acc[ti][tj][r];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ti * 32 + crow_g(r, half), col = tj * 32 + l31;
          float v = acc[ti][tj][r];
          v += s_part[0][row * kGT + col];
          v += s_part[1][row * kGT + col];
          v += s_part[2][row * kGT + col];
          pr.out[static_cast<size_t>(i0 + row) * pr.ldout + j0 + col] = v;
        }
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_grouped_gemm_tn_f32(const CodaTnProblem *problems, int count, void *stream) {
  using namespace coda;
  if (count < 0) return CODA_EINVAL;
  if (count == 0) return CODA_OK;
  if (!problems) return CODA_EINVAL;
  for (int i = 0; i < count; ++i) {
    const CodaTnProblem &q = problems[i];
    if (!q.dy || !q.x || !q.out || q.rows <= 0 || q.m <= 0 || q.n <= 0) return CODA_EINVAL;
    if (q.rows % 8 || q.m % kGT || q.n % kGT || q.m / kGT > 65535 || q.n / kGT > 65535) return CODA_ENOSPC;
    if (q.lddy < q.m || q.ldx < q.n || q.ldout < q.n) return CODA_EINVAL;
  }
  clear_sticky_error();
  for (int first = 0; first < count; first += kMaxProblems) {
    Batch b;
    b.count = min(kMaxProblems, count - first);
    int slots = 0;
    for (int i = 0; i < b.count; ++i) {
      const CodaTnProblem &q = problems[first + i];
      Problem &p = b.p[i];
      p.dy = q.dy;
      p.x = q.x;
      p.out = q.out;
      p.rows = q.rows;
      p.lddy = static_cast<int>(q.lddy);
      p.ldx = static_cast<int>(q.ldx);
      p.ldout = static_cast<int>(q.ldout);
      p.tiles_m = static_cast<unsigned short>(q.m / kGT);
      p.tiles_n = static_cast<unsigned short>(q.n / kGT);
      p.pad_ = 0;
      slots = max(slots, static_cast<int>(p.tiles_m) * p.tiles_n);
    }
    b.slots_per_problem = slots;
    const int groups = (b.count + kXcds - 1) / kXcds;  // problems per XCD
    hipLaunchKernelGGL(grouped_tn_kernel, dim3(static_cast<unsigned>(groups) * slots * kXcds), dim3(kGThreads), 0,
                       static_cast<hipStream_t>(stream), b);
    const int st = launch_status();
    if (st != CODA_OK) return st;
  }
  return CODA_OK;
}
