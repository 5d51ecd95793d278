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
// operands are used exactly as they lie in memory: A[i][k-slot] = dy[k][column of i] and B[k-slot][j] = x[k][column
// of j] are contiguous per k-slot, so lanes load their operand values straight from global memory (L2) -- no LDS
// staging, no transposition.  Tiles of one problem are dealt to ONE XCD (workgroup g runs on XCD g % 8), so a
// problem's operands are fetched from HBM once (FETCH_SIZE = 1.02 x the operand bytes) and re-read by its other
// tiles from that XCD's L2.
//
// What bounds it (probe variants, 72 problems of 2048 x 256 x 256): MFMAs alone 166 us, loads alone 118 us, and
// the first version -- one dword per lane, operand blocks loaded in bursts -- took their SUM and more (341 us):
// the vector-memory pipe takes ~16 cycles per wave instruction whatever its width, and bursts from the eight
// waves of a CU queued behind each other while their MFMA pipes idled.  8-byte loads (two adjacent columns per
// lane, component c feeding MFMA tile c) and a software pipeline that issues one operand pair per k-step,
// 16 steps ahead of its use, bring it to 234 us = 83 TFLOP/s (library, one call per product: 1080 us).
#include "coda_gemm.h"
#include "common.hip.h"

#include <cstdint>

// end of a load / MFMA phase: nothing moves across, neither in the IR (the memory clobber keeps loads from
// being sunk next to their uses) nor in the machine scheduler
#define CODA_PHASE_FENCE()            \
  do {                                \
    asm volatile("" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kGT = 64;        // tile edge
constexpr int kGThreads = 256;
constexpr int kXcds = 8;

struct Problem {
  const float *dy, *x;
  float *out;
  int rows, lddy, ldx, ldout;
  unsigned short tiles_m, tiles_n;  // 64-wide tiles
  int pad_;
};
constexpr int kMaxProblems = 80;  // 48 B each: 3.8 KB of kernel arguments (the limit is 4 KB)
struct Batch {
  Problem p[kMaxProblems];
  int count;
  int slots_per_problem;  // tiles of the largest problem in the batch
};

__device__ __forceinline__ int crow_g(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// kUnroll: k-steps (of 2 rows) between the load of an operand pair and its use (= loads in flight per wave / 2)
template <int kUnroll, int kOcc>
__global__ __launch_bounds__(kGThreads, kOcc) void grouped_tn_kernel(const Batch batch) {
  __shared__ __attribute__((aligned(16))) float s_part[2][kGT * kGT];  // partial tiles, two reduction rounds
  // XCD-aware dealing: workgroup g -> XCD g % 8; a problem's tiles share an XCD
  const int g = blockIdx.x, xcd = g % kXcds, slot = g / kXcds;
  const int pidx = (slot / batch.slots_per_problem) * kXcds + xcd, tile = slot % batch.slots_per_problem;
  if (pidx >= batch.count) return;
  const Problem &pr = batch.p[pidx];
  if (tile >= pr.tiles_m * pr.tiles_n) return;
  const int i0 = (tile / pr.tiles_n) * kGT, j0 = (tile % pr.tiles_n) * kGT;
  const int lane = lane_id(), w = wave_id(), half = lane >> 5, l31 = lane & 31;
  const int per_wave = pr.rows / 4;  // rows is a multiple of 8 (host check): an even count per wave
  // A lane fetches TWO adjacent columns (one 8-byte load) of the row its k-slot names: the vector-memory pipe
  // takes ~16 cycles per wave instruction whatever its width, and with one dword per lane that alone cost as much
  // as the MFMAs.  Component c of the pair feeds MFMA tile c, so tile (ta, tb) holds C[i0 + 2 i + ta][j0 + 2 j + tb].
  const float *pa = pr.dy + static_cast<size_t>(w * per_wave + half) * pr.lddy + i0 + 2 * l31;
  const float *pb = pr.x + static_cast<size_t>(w * per_wave + half) * pr.ldx + j0 + 2 * l31;
  const size_t sa = 2 * static_cast<size_t>(pr.lddy), sb = 2 * static_cast<size_t>(pr.ldx);
  f32x16 acc[2][2] = {};
  const int steps = per_wave / 2;
  auto ld2 = [](const float *p) { return *reinterpret_cast<const float2 *>(p); };
  auto mma = [&](float2 a, float2 b) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
  };
  // Software pipeline at k-step granularity: slot u of the ring holds the operands of a k-step kUnroll steps
  // ahead; every iteration issues ONE pair of loads and four MFMAs, so the vector-memory pipe sees an even
  // stream instead of bursts (bursts of all eight waves of a CU serialised behind each other).  The fences keep
  // both the IR and the machine scheduler from regrouping loads and uses.
  float2 ra[kUnroll], rb[kUnroll];
  const int nfull = steps / kUnroll;  // ring revolutions
  const float *qa = pa, *qb = pb;
  if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      ra[u] = ld2(qa + u * sa);
      rb[u] = ld2(qb + u * sb);
    }
    qa += kUnroll * sa;
    qb += kUnroll * sb;
  }
  for (int rev = 0; rev < nfull; ++rev) {
    // the last revolution re-reads the wave's first rows (unconditional loads keep the wait counts exact)
    const bool more = rev + 1 < nfull;
    const float *na = more ? qa : pa, *nb = more ? qb : pb;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const float2 a = ra[u], b = rb[u];
      CODA_PHASE_FENCE();
      ra[u] = ld2(na + u * sa);
      rb[u] = ld2(nb + u * sb);
      CODA_PHASE_FENCE();
      mma(a, b);
    }
    qa += more ? kUnroll * sa : 0;
    qb += more ? kUnroll * sb : 0;
  }
  for (int s = nfull * kUnroll; s < steps; ++s) {
    mma(ld2(qa), ld2(qb));
    qa += sa;
    qb += sb;
  }
  // combine in two rounds through 32 KB of LDS, fixed order (w0 + w2) + (w1 + w3): waves 2, 3 park their tiles,
  // waves 0, 1 add them; then wave 1 parks its sum and wave 0 adds it and stores
  auto park = [&](float *dst) {
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float2 *>(&dst[(2 * crow_g(r, half) + ta) * kGT + 2 * l31]) =
            make_float2(acc[ta][0][r], acc[ta][1][r]);
  };
  auto add_from = [&](const float *src) {
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 t = *reinterpret_cast<const float2 *>(&src[(2 * crow_g(r, half) + ta) * kGT + 2 * l31]);
        acc[ta][0][r] += t.x;
        acc[ta][1][r] += t.y;
      }
  };
  if (w >= 2) park(s_part[w - 2]);
  __syncthreads();
  if (w < 2) add_from(s_part[w]);
  __syncthreads();
  if (w == 1) park(s_part[0]);
  __syncthreads();
  if (w == 0) {
    add_from(s_part[0]);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 2 * crow_g(r, half) + ta;
        *reinterpret_cast<float2 *>(&pr.out[static_cast<size_t>(i0 + row) * pr.ldout + j0 + 2 * l31]) =
            make_float2(acc[ta][0][r], acc[ta][1][r]);
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
    // 8-byte operand / result accesses
    if (q.lddy % 2 || q.ldx % 2 || q.ldout % 2 || (reinterpret_cast<uintptr_t>(q.dy) | reinterpret_cast<uintptr_t>(q.x) |
                                                     reinterpret_cast<uintptr_t>(q.out)) % 8)
      return CODA_ENOSPC;
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
    // 16 k-steps (32 operand loads) in flight per wave: measured 234 us for the decoder's 72 problems against
    // 246 us with 8, 263 us with 4 (tools/bench_grouped_tn.py)
    hipLaunchKernelGGL((grouped_tn_kernel<16, 2>), dim3(static_cast<unsigned>(groups) * slots * kXcds), dim3(kGThreads), 0,
                       static_cast<hipStream_t>(stream), b);
    const int st = launch_status();
    if (st != CODA_OK) return st;
  }
  return CODA_OK;
}
