// gemm_nn.hip -- own fp32-MFMA GEMM for the token-wise linear layers of the transformer stacks:
//   C (M x N) [+]= A (M x K) . op(B) [+ bias],   op(B) = B^T with B (N x K)   [forward:  y = x W^T + b]
//                                                 op(B) = B   with B (K x N)   [backward: dx = dy W]
// (the third form, dW = dy^T x, is gemm_tn.hip).
//
// Why not the library: the decoder issues ~230 such products per training step at 2048 x 256 x 256, where
// hipBLASLt spends 8-15 us on the GPU and ~14 us of HOST time per call (tools/host_profile.py) -- the step is
// launch-bound on the host.  This kernel is one plain launch (~3 us of host time through the C ABI).
//
// Shape: a workgroup of 4 waves owns a 64 x 64 tile of C (wave = one 32 x 32 quadrant, 16 accumulator
// registers), K is walked in steps of 32 through LDS with a register prefetch of the next step (one barrier per
// step, two LDS buffers).  v_mfma_f32_32x32x2_f32 contracts two k per instruction, A[i = lane&31][k-slot = lane>>5]:
// slot h of instruction s is k = 16 h + s, so a lane's sixteen A values are 64 contiguous bytes of one row
// (4 x ds_read_b128, rows padded to 36 floats: conflict-free).  B^T tiles are staged the same way; plain-B tiles
// stay [k][n] and are read with ds_read_b32 (lanes = consecutive n).  fp32 MFMA issues one instruction per 64
// cycles and SIMD, so the 4-8 LDS reads per 16 instructions hide completely; the kernel is MFMA-issue bound.
#include "coda_gemm.h"
#include "common.hip.h"
#include "dropout.hip.h"

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTile = 64, kStep = 32, kPad = kStep + 4, kThreadsNN = 256;

__device__ __forceinline__ int crow_nn(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <bool BT>
__global__ __launch_bounds__(kThreadsNN) void sgemm_kernel(const float *__restrict__ a, long long lda,
                                                          const float *__restrict__ b, long long ldb,
                                                          float *__restrict__ c, long long ldc,
                                                          const float *__restrict__ bias, int n_tiles, int k,
                                                          int accumulate) {
  // A: [2][64][36]; B: BT ? [2][64][36] : [2][32][64]
  __shared__ __attribute__((aligned(16))) float s_a[2][kTile * kPad];
  __shared__ __attribute__((aligned(16))) float s_b[2][BT ? kTile * kPad : kStep * kTile];
  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = (static_cast<int>(blockIdx.x) / n_tiles) * kTile, n0 = (static_cast<int>(blockIdx.x) % n_tiles) * kTile;
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32;

  // this thread's two 16-B pieces of each staged tile
  float4 ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * kThreadsNN;
      ra[u] = *reinterpret_cast<const float4 *>(a + static_cast<size_t>(m0 + (idx >> 3)) * lda + k0 + 4 * (idx & 7));
      if (BT) rb[u] = *reinterpret_cast<const float4 *>(b + static_cast<size_t>(n0 + (idx >> 3)) * ldb + k0 + 4 * (idx & 7));
      else rb[u] = *reinterpret_cast<const float4 *>(b + static_cast<size_t>(k0 + (idx >> 4)) * ldb + n0 + 4 * (idx & 15));
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * kThreadsNN;
      *reinterpret_cast<float4 *>(&s_a[buf][(idx >> 3) * kPad + 4 * (idx & 7)]) = ra[u];
      if (BT) *reinterpret_cast<float4 *>(&s_b[buf][(idx >> 3) * kPad + 4 * (idx & 7)]) = rb[u];
      else *reinterpret_cast<float4 *>(&s_b[buf][(idx >> 4) * kTile + 4 * (idx & 15)]) = rb[u];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < k; k0 += kStep, buf ^= 1) {
    const bool more = k0 + kStep < k;
    if (more) fetch(k0 + kStep);
    const float *ta = &s_a[buf][(wm + l31) * kPad + 16 * half];
    float av[16], bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(ta + 4 * q);
      av[4 * q] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
    }
    if (BT) {
      const float *tb = &s_b[buf][(wn + l31) * kPad + 16 * half];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(tb + 4 * q);
        bv[4 * q] = t.x; bv[4 * q + 1] = t.y; bv[4 * q + 2] = t.z; bv[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int sidx = 0; sidx < 16; ++sidx) bv[sidx] = s_b[buf][(16 * half + sidx) * kTile + wn + l31];
    }
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sidx], bv[sidx], acc, 0, 0, 0);
    if (more) stage(buf ^ 1);
    __syncthreads();
  }
  // acc[r]: row m0 + wm + crow(r, half), column n0 + wn + l31
  const int col = n0 + wn + l31;
  const float bval = bias ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float *p = c + static_cast<size_t>(m0 + wm + crow_nn(r, half)) * ldc + col;
    float v = acc[r] + bval;
    if (accumulate) v += *p;
    *p = v;
  }
}

// Launch-sized problems (2048 x 256 x 256: 0.27 GFLOP): a 64 x 64 tiling gives 128 workgroups whose waves each
// issue K/2 dependent MFMAs back to back (8 us).  Here a workgroup owns a 32 x 32 tile of C and its four waves
// split K: 512 workgroups, 32 MFMAs per wave, operands straight from global memory into registers (a lane reads 64
// contiguous bytes of its row; no LDS staging, no barrier in the loop), partial tiles summed through LDS.
// EPI = 1: C = dropout(relu(A . op(B) + bias)) -- the feed-forward's first layer with the element-wise pass that
// followed it (token_ln.hip: bias_relu_dropout_fwd_kernel) in the epilogue; same counter hash of (seed, row * n + col),
// so the masks are those of the stand-alone kernel.
// EPI = 2: C = relu-dropout BACKWARD of the product, dz = (act > 0 ? A . op(B) * inv_keep : 0) with `act` the saved
// activation (token_ln.hip: bias_relu_dropout_bwd_kernel, a dropped or clamped element has act == 0 either way), plus the
// column sums of dz over the workgroup's 32 rows -> partials[m0 / 32][col] (the bias gradient's per-block partials: the
// caller reduces the m / 32 rows in fixed order, coda_tok_colsum_finalize_grouped_f32).
struct ReluDrop {
  uint32_t thresh24, seed;
  float inv_keep;
  int n;
  const float *act = nullptr;
  float *partials = nullptr;
};
template <bool BT, int EPI = 0>
__global__ __launch_bounds__(kThreadsNN) void sgemm_splitk_kernel(const float *__restrict__ a, long long lda,
                                                                 const float *__restrict__ b, long long ldb,
                                                                 float *__restrict__ c, long long ldc,
                                                                 const float *__restrict__ bias, int n_tiles, int k,
                                                                 int accumulate, ReluDrop epi = ReluDrop{0u, 0u, 1.f, 0}) {
  __shared__ float s_part[4][16][kWave];
  const int lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = (static_cast<int>(blockIdx.x) / n_tiles) * 32, n0 = (static_cast<int>(blockIdx.x) % n_tiles) * 32;
  const int kc = k / 4, kw = w * kc;  // this wave's K range (a multiple of 32)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float *arow = a + static_cast<size_t>(m0 + l31) * lda + kw + 16 * half;
  const float *brow = BT ? b + static_cast<size_t>(n0 + l31) * ldb + kw + 16 * half
                         : b + static_cast<size_t>(kw + 16 * half) * ldb + n0 + l31;
  // Two register sets, two steps per pass: the loads of both steps are issued together and the 16 MFMAs of the first
  // run under the second's (a wave has only K / 128 steps -- two at K = 256 -- and every step used to pay a full
  // memory round trip before its MFMAs).
  float av[2][16], bv[2][16];
  auto load = [&](int set, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(arow + k0 + 4 * q);
      av[set][4 * q] = t.x; av[set][4 * q + 1] = t.y; av[set][4 * q + 2] = t.z; av[set][4 * q + 3] = t.w;
    }
    if (BT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(brow + k0 + 4 * q);
        bv[set][4 * q] = t.x; bv[set][4 * q + 1] = t.y; bv[set][4 * q + 2] = t.z; bv[set][4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int sidx = 0; sidx < 16; ++sidx) bv[set][sidx] = brow[static_cast<size_t>(k0 + sidx) * ldb];
    }
  };
  auto mfma = [&](int set) {
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][sidx], bv[set][sidx], acc, 0, 0, 0);
  };
  int k0 = 0;
  for (; k0 + 2 * kStep <= kc; k0 += 2 * kStep) {  // (no branch between the loads and the MFMAs: partial vmcnt waits)
    load(0, k0);
    load(1, k0 + kStep);
    __builtin_amdgcn_sched_barrier(0);  // (else the scheduler sinks every load next to its MFMA: one round trip each)
    mfma(0);
    mfma(1);
  }
  if (k0 < kc) {
    load(0, k0);
    __builtin_amdgcn_sched_barrier(0);
    mfma(0);
  }
  // Every wave posts its partial tile; wave w then finishes registers 4 w .. 4 w + 3 (rows 8 w + 4 half .. + 3 of the tile):
  // the epilogue -- bias, activation (backward), 16 row stores -- used to be wave 0's alone, a quarter of the launch's
  // duration at 2048 x 256 x 256, with the other three waves gone.  Same order of additions as before.
#pragma unroll
  for (int r = 0; r < 16; ++r) s_part[w][r][lane] = acc[r];
  __syncthreads();
  const int col = n0 + l31;
  const float bval = bias ? bias[col] : 0.f;
  float colsum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * w + q;
    float *p = c + static_cast<size_t>(m0 + crow_nn(r, half)) * ldc + col;
    float v = ((s_part[0][r][lane] + s_part[1][r][lane]) + (s_part[2][r][lane] + s_part[3][r][lane])) + bval;
    if (accumulate) v += *p;
    if (EPI == 2) {
      const float av = epi.act[static_cast<size_t>(m0 + crow_nn(r, half)) * epi.n + col];
      v = av > 0.f ? v * epi.inv_keep : 0.f;
      colsum += v;
    }
    if (EPI == 1) {
      v = fmaxf(v, 0.f);
      if (epi.thresh24) {
        const uint32_t idx = static_cast<uint32_t>(m0 + crow_nn(r, half)) * static_cast<uint32_t>(epi.n) + static_cast<uint32_t>(col);
        v = keep_elem(epi.seed, idx, epi.thresh24) ? v * epi.inv_keep : 0.f;
      }
    }
    *p = v;
  }
  if (EPI == 2) {  // the tile's column sums: the waves' 8-row sums meet in LDS (fixed order)
    colsum += __shfl_xor(colsum, 32, kWave);
    __syncthreads();  // (every wave has read what it needed of s_part)
    if (half == 0) s_part[0][w][l31] = colsum;
    __syncthreads();
    if (w == 0 && half == 0)
      epi.partials[static_cast<size_t>(m0 / 32) * epi.n + col] =
          (s_part[0][0][l31] + s_part[0][1][l31]) + (s_part[0][2][l31] + s_part[0][3][l31]);
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_sgemm_f32(int transb, int m, int n, int k, const float *a, long long lda, const float *b,
                            long long ldb, float *c, long long ldc, const float *bias, int accumulate, void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (!a || !b || !c || k == 0) return CODA_EINVAL;
  // tile constraints (callers route other shapes to coda_gemm_f32): CODA_ENOSPC = "not this kernel's shape"
  if (m % kTile || n % kTile || k % kStep || lda % 4 || ldb % 4 || (reinterpret_cast<uintptr_t>(a) & 15) ||
      (reinterpret_cast<uintptr_t>(b) & 15))
    return CODA_ENOSPC;
  if (lda < k || ldb < (transb ? k : n) || ldc < n) return CODA_EINVAL;
  clear_sticky_error();
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (static_cast<long long>(m) * n <= 2048ll * 1024 && k % 128 == 0) {  // launch-sized: split-K over the waves
    const int nt = n / 32;
    const dim3 sgrid(static_cast<unsigned>((m / 32) * nt));
    if (transb) hipLaunchKernelGGL(sgemm_splitk_kernel<true>, sgrid, dim3(kThreadsNN), 0, s, a, lda, b, ldb, c, ldc, bias, nt, k, accumulate);
    else hipLaunchKernelGGL(sgemm_splitk_kernel<false>, sgrid, dim3(kThreadsNN), 0, s, a, lda, b, ldb, c, ldc, bias, nt, k, accumulate);
    return launch_status();
  }
  const int n_tiles = n / kTile;
  const dim3 grid(static_cast<unsigned>((m / kTile) * n_tiles));
  if (transb) hipLaunchKernelGGL(sgemm_kernel<true>, grid, dim3(kThreadsNN), 0, s, a, lda, b, ldb, c, ldc, bias, n_tiles, k, accumulate);
  else hipLaunchKernelGGL(sgemm_kernel<false>, grid, dim3(kThreadsNN), 0, s, a, lda, b, ldb, c, ldc, bias, n_tiles, k, accumulate);
  return launch_status();
}

CODA_API int coda_sgemm_relu_dropout_f32(int transb, int m, int n, int k, const float *a, long long lda, const float *b,
                                         long long ldb, float *c, long long ldc, const float *bias, float dropout_p,
                                         uint64_t seed, void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0 || !(dropout_p >= 0.f) || dropout_p >= 1.f) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (!a || !b || !c || k == 0) return CODA_EINVAL;
  if (lda < k || ldb < (transb ? k : n)) return CODA_EINVAL;  // same operand checks as coda_sgemm_f32
  // the launch-sized split-K kernel only (a wave-0 epilogue owns whole sums there); anything else: separate passes
  if (m % 64 || n % 64 || k % 128 || (lda | ldb) % 4 || ldc != n || (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 ||
      static_cast<long long>(m) * n > 2048ll * 1024)
    return CODA_ENOSPC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  ReluDrop epi{drop_thresh24(dropout_p), static_cast<uint32_t>(seed ^ (seed >> 32)), 1.0f / (1.0f - dropout_p), n};
  const int nt = n / 32;
  const dim3 sgrid(static_cast<unsigned>((m / 32) * nt));
  if (transb) hipLaunchKernelGGL((sgemm_splitk_kernel<true, 1>), sgrid, dim3(kThreadsNN), 0, s, a, lda, b, ldb, c, ldc, bias, nt, k, 0, epi);
  else hipLaunchKernelGGL((sgemm_splitk_kernel<false, 1>), sgrid, dim3(kThreadsNN), 0, s, a, lda, b, ldb, c, ldc, bias, nt, k, 0, epi);
  return launch_status();
}

CODA_API int coda_sgemm_relu_dropout_bwd_blocks(int m) { return m > 0 ? m / 32 : 0; }

CODA_API int coda_sgemm_relu_dropout_bwd_f32(int m, int n, int k, const float *da, long long ldda, const float *w,
                                             long long ldw, const float *act, float dropout_p, float *dz,
                                             float *partials, void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0 || !(dropout_p >= 0.f) || dropout_p >= 1.f) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (!da || !w || !act || !dz || !partials || k == 0) return CODA_EINVAL;
  if (ldda < k || ldw < n) return CODA_EINVAL;
  // the launch-sized split-K kernel only (wave 0's epilogue owns whole sums and the tile's 32 rows)
  if (m % 64 || n % 64 || k % 128 || (ldda | ldw) % 4 || (reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(w)) % 16 ||
      static_cast<long long>(m) * n > 2048ll * 1024)
    return CODA_ENOSPC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  ReluDrop epi{0u, 0u, 1.0f / (1.0f - dropout_p), n, act, partials};
  const int nt = n / 32;
  const dim3 sgrid(static_cast<unsigned>((m / 32) * nt));
  hipLaunchKernelGGL((sgemm_splitk_kernel<false, 2>), sgrid, dim3(kThreadsNN), 0, s, da, ldda, w, ldw, dz, static_cast<long long>(n),
                     nullptr, nt, k, 0, epi);
  return launch_status();
}
