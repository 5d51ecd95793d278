// gemm_x3.hip -- the dense projections of the transformer stacks (y = x W^T + b, dx = dy W) as fp32-ACCURATE GEMMs on the
// bf16 matrix cores.
//
// Why: the fp32 "MFMA" of gfx950 runs at the fp32 VECTOR rate (157 TFLOP/s, and it does not overlap VALU work:
// tools/mfma_valu_probe.hip), the bf16 one at 2.5 PFLOP/s on matrix cores of its own.  An fp32 value is EXACTLY the sum
// of three bf16 pieces, x = hi + mid + lo (8 + 8 + 8 significand bits, the two residuals are exact fp32 subtractions),
// and a product a.b is then the sum of nine piece products, each exact in fp32.  This file keeps the SIX of order <= 2
// (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) and drops mid.lo, lo.mid, lo.lo: together <= 2^-24 |a.b|, half of
// the rounding an fp32 FMA makes on the same product.  Accumulation is the MFMA's fp32 accumulator.  Measured against
// float64 the result is within 2x of the library's native fp32 GEMM on well-scaled, wide-range and cancelling operands
// (tests/test_gemm_x3_gpu.py).  Six v_mfma_f32_32x32x16_bf16 cost 6/16 of the fp32 MFMAs they replace.
//
// Round 2 had a first version (nine products, BOTH operands split per staged tile, measured 1.04-1.16x the library and
// removed in round 4).  What is different here:
//   * the WEIGHT operand is split ONCE per optimizer step by a small kernel of its own (coda_gemm_x3_split_f32: all
//     weights of a step in one launch, both orientations -- [N][K] planes for y = x W^T and [K][N] planes for dx = dy W,
//     so both products are the same "NT" kernel and nothing is ever transposed on the way into LDS); the GEMM kernel
//     stages ready-made bf16 tiles for it (no VALU, 16-byte LDS writes);
//   * only the ACTIVATION operand (streamed once, M = 16 384 token rows) is split in the kernel, once per staged tile:
//     ~6 VALU instructions per element amortised over the tile's 64-128 output columns, issued next to MFMAs that --
//     unlike the fp32 ones -- leave the vector pipe free;
//   * six products instead of nine, a workgroup tile mapping that keeps an A row block on one XCD's L2.
//
//   C (M x N, row stride ldc) [+]= A (M x K fp32, row stride lda) . W^T [+ bias],   W as three bf16 planes [3][N][K]
//
// Shape: 256 threads = 2 x 2 waves on a BM x BN tile of C (128 x 128, 64 x 128, 64 x 64), K walked in steps of 32 through
// one LDS stage ([row][32 k] bf16 per plane, row stride 80 B: conflict-free ds_read_b128 fragments), the global loads of
// step i + 1 in flight during the MFMAs of step i, two workgroups per CU.
#include "coda_gemm.h"
#include "common.hip.h"

#include <cstdint>

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

constexpr int kX3Threads = 256;
constexpr int kX3BK = 32;        // k per stage
constexpr int kX3Row = 80;       // bytes per LDS row: 32 bf16 + 16 B pad

struct X3NtParams {
  const float *a;
  const __bf16 *w;     // plane q at w + q * wplane, row r at + r * ldw (elements)
  const float *bias;
  float *c;
  long long lda, ldw, wplane, ldc;
  int m, n, k;
  int accumulate;
  int tiles_m, tiles_n;
  int xcd_map;         // tiles of one A row block on one XCD
};

struct Pieces4 {
  bf16x4 p[3];
};
// x = hi + mid + lo exactly (round-to-nearest-even conversions; both residuals are exact in fp32)
__device__ __forceinline__ Pieces4 split4(f32x4v x) {
  Pieces4 r;
  r.p[0] = __builtin_convertvector(x, bf16x4);
  x = x - __builtin_convertvector(r.p[0], f32x4v);
  r.p[1] = __builtin_convertvector(x, bf16x4);
  x = x - __builtin_convertvector(r.p[1], f32x4v);
  r.p[2] = __builtin_convertvector(x, bf16x4);
  return r;
}

__device__ __forceinline__ int crow_x3(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int BM, int BN>
__global__ __launch_bounds__(kX3Threads, 2) void x3_nt_kernel(const X3NtParams p) {
  constexpr int TM = BM / 64, TN = BN / 64;          // 32 x 32 MFMA tiles per wave in each direction
  constexpr int IMG_A = BM * kX3Row, IMG_W = BN * kX3Row;
  constexpr int NA = BM * kX3BK / 4 / kX3Threads;    // float4 pieces of A per thread and stage
  constexpr int NW = 3 * BN * 4 / kX3Threads;        // 16-byte pieces of the W planes per thread and stage
  static_assert(NA >= 1 && NW >= 1 && 3 * BN * 4 % kX3Threads == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * IMG_A + 3 * IMG_W];
  unsigned char *s_a = smem, *s_w = smem + 3 * IMG_A;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  int tm, tn;
  {
    const int b = static_cast<int>(blockIdx.x);
    if (p.xcd_map) {
      // workgroups are dealt round-robin over the 8 XCDs: the tiles_n tiles that share an A row block take
      // consecutive slots of ONE XCD, so the block is fetched from HBM once and served from that L2 afterwards
      const int xcd = b & 7, slot = b >> 3;
      tm = (slot / p.tiles_n) * 8 + xcd;
      tn = slot % p.tiles_n;
    } else {
      tm = b / p.tiles_n;
      tn = b % p.tiles_n;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (no lambdas around the staging registers: captured by reference, hipcc kept `rw` in SCRATCH memory -- every
  // prefetched piece was waited for at once, stored to scratch and read back before its LDS write)
  f32x4v ra[NA];
  u32x4v rw[NW];
#define X3_FETCH(K0)                                                                                                   \
  {                                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < NA; ++u) {                                                                   \
      const int i_ = tid + u * kX3Threads, row_ = i_ >> 3, c4_ = i_ & 7;                                               \
      ra[u] = *reinterpret_cast<const f32x4v *>(p.a + static_cast<size_t>(m0 + row_) * p.lda + (K0) + 4 * c4_);        \
    }                                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < NW; ++u) {                                                                   \
      const int i_ = tid + u * kX3Threads, q_ = i_ / (BN * 4), rem_ = i_ % (BN * 4), row_ = rem_ >> 2, c_ = rem_ & 3;  \
      rw[u] = *reinterpret_cast<const u32x4v *>(p.w + static_cast<size_t>(q_) * p.wplane +                              \
                                               static_cast<size_t>(n0 + row_) * p.ldw + (K0) + 8 * c_);                \
    }                                                                                                                  \
  }

  X3_FETCH(0);
  for (int k0 = 0; k0 < p.k; k0 += kX3BK) {
    __syncthreads();  // everyone is done reading the previous stage
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int i = tid + u * kX3Threads, row = i >> 3, c4 = i & 7;
      const Pieces4 sp = split4(ra[u]);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4 *>(s_a + q * IMG_A + row * kX3Row + 8 * c4) = sp.p[q];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = tid + u * kX3Threads, q = i / (BN * 4), rem = i % (BN * 4), row = rem >> 2, c = rem & 3;
      *reinterpret_cast<u32x4v *>(s_w + q * IMG_W + row * kX3Row + 16 * c) = rw[u];
    }
    __syncthreads();
    {
      // the next stage's loads, in flight during this stage's MFMAs (the last iteration fetches its own tile again:
      // unconditional loads keep the waits in front of the LDS writes exact)
      const int kn = k0 + kX3BK < p.k ? k0 + kX3BK : k0;
      X3_FETCH(kn);
    }
#pragma unroll
    for (int kk = 0; kk < kX3BK / 16; ++kk) {
      bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          fa[i][q] = *reinterpret_cast<const bf16x8 *>(s_a + q * IMG_A + (wm + 32 * i + l31) * kX3Row + 32 * kk + 16 * half);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          fb[j][q] = *reinterpret_cast<const bf16x8 *>(s_w + q * IMG_W + (wn + 32 * j + l31) * kX3Row + 32 * kk + 16 * half);
      // six piece products of order <= 2, smallest first; the (i, j) tiles interleave so that consecutive MFMAs are
      // independent
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int qa_of[6] = {2, 0, 1, 1, 0, 0};
        constexpr int qb_of[6] = {0, 2, 1, 0, 1, 0};
        const int qa = qa_of[t], qb = qb_of[t];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acc[i][j], 0, 0, 0);
      }
    }
  }
#undef X3_FETCH
  // epilogue: a store instruction writes 2 rows x 128 contiguous bytes
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + l31;
      const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + crow_x3(r, half);
        float *dst = p.c + static_cast<size_t>(row) * p.ldc + col;
        float v = acc[i][j][r] + bias;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
}

// ---- the weights' pieces: every weight of a step in one launch ---------------------------------------------------------
constexpr int kSplitMaxItems = 48;
struct X3SplitTable {
  const float *src[kSplitMaxItems];
  __bf16 *nt[kSplitMaxItems];   // [3][rows][cols] (may be null)
  __bf16 *nn[kSplitMaxItems];   // [3][cols][rows] (may be null)
  long long ld[kSplitMaxItems];
  int rows[kSplitMaxItems], cols[kSplitMaxItems];
  int first_tile[kSplitMaxItems + 1];
  int count;
};

// one workgroup per 32 x 32 tile of one weight
__global__ __launch_bounds__(256) void x3_split_kernel(const X3SplitTable t) {
  __shared__ float s[32][33];
  const int b = static_cast<int>(blockIdx.x);
  int it = 0;
  while (it + 1 < t.count && b >= t.first_tile[it + 1]) ++it;
  const int rows = t.rows[it], cols = t.cols[it];
  const int tiles_c = (cols + 31) / 32, tile = b - t.first_tile[it];
  const int r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows per pass
  const float *src = t.src[it];
  const long long ld = t.ld[it];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + ty + 8 * q, c = c0 + tx;
    s[ty + 8 * q][tx] = (r < rows && c < cols) ? src[static_cast<size_t>(r) * ld + c] : 0.f;
  }
  __syncthreads();
  auto pieces = [](float x, __bf16 (&o)[3]) {
    o[0] = static_cast<__bf16>(x);
    x -= static_cast<float>(o[0]);
    o[1] = static_cast<__bf16>(x);
    x -= static_cast<float>(o[1]);
    o[2] = static_cast<__bf16>(x);
  };
  const size_t plane = static_cast<size_t>(rows) * cols;
  if (t.nt[it]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + ty + 8 * q, c = c0 + tx;
      if (r < rows && c < cols) {
        __bf16 o[3];
        pieces(s[ty + 8 * q][tx], o);
#pragma unroll
        for (int k = 0; k < 3; ++k) t.nt[it][k * plane + static_cast<size_t>(r) * cols + c] = o[k];
      }
    }
  }
  if (t.nn[it]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + ty + 8 * q, r = r0 + tx;  // transposed walk: consecutive lanes = consecutive rows of the source
      if (r < rows && c < cols) {
        __bf16 o[3];
        pieces(s[tx][ty + 8 * q], o);
#pragma unroll
        for (int k = 0; k < 3; ++k) t.nn[it][k * plane + static_cast<size_t>(c) * rows + r] = o[k];
      }
    }
  }
}

template <int BM, int BN>
int launch_x3_nt(X3NtParams p, hipStream_t s) {
  p.tiles_m = p.m / BM;
  p.tiles_n = p.n / BN;
  p.xcd_map = (p.tiles_m % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((x3_nt_kernel<BM, BN>), dim3(static_cast<unsigned>(p.tiles_m) * p.tiles_n), dim3(kX3Threads), 0, s, p);
  return launch_status();
}

}  // namespace
}  // namespace coda

CODA_API int coda_gemm_x3_split_f32(const CodaX3SplitItem *items, int count, void *stream) {
  using namespace coda;
  if (count < 0 || (count > 0 && !items)) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  for (int base = 0; base < count; base += kSplitMaxItems) {
    X3SplitTable t;
    const int nb = count - base < kSplitMaxItems ? count - base : kSplitMaxItems;
    int tiles = 0;
    for (int i = 0; i < nb; ++i) {
      const CodaX3SplitItem &it = items[base + i];
      if (!it.src || it.rows <= 0 || it.cols <= 0 || it.ld < it.cols || (!it.nt && !it.nn)) return CODA_EINVAL;
      t.src[i] = it.src;
      t.nt[i] = static_cast<__bf16 *>(it.nt);
      t.nn[i] = static_cast<__bf16 *>(it.nn);
      t.ld[i] = it.ld;
      t.rows[i] = it.rows;
      t.cols[i] = it.cols;
      t.first_tile[i] = tiles;
      tiles += ((it.rows + 31) / 32) * ((it.cols + 31) / 32);
    }
    t.first_tile[nb] = tiles;
    t.count = nb;
    hipLaunchKernelGGL(x3_split_kernel, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, s, t);
  }
  return launch_status();
}

CODA_API int coda_gemm_x3_nt_f32(int m, int n, int k, const float *a, long long lda, const void *w_planes, long long ldw,
                                 long long plane_stride, float *c, long long ldc, const float *bias, int accumulate,
                                 void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (!a || !w_planes || !c || k == 0) return CODA_EINVAL;
  // shapes / alignments this kernel takes (everything else: coda_gemm_f32)
  if (m % 64 || n % 64 || k % kX3BK || lda % 4 || ldw % 8 || plane_stride % 8 ||
      (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w_planes)) % 16 ||
      reinterpret_cast<uintptr_t>(c) % 4)
    return CODA_ENOSPC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  X3NtParams p{a, static_cast<const __bf16 *>(w_planes), bias, c, lda, ldw, plane_stride, ldc, m, n, k, accumulate, 0, 0, 0};
  // tile choice: the largest tile that still gives every CU two workgroups
  const long long t128 = (m % 128 == 0 && n % 128 == 0) ? static_cast<long long>(m / 128) * (n / 128) : 0;
  const long long t64x128 = (n % 128 == 0) ? static_cast<long long>(m / 64) * (n / 128) : 0;
  if (t128 >= 512) return launch_x3_nt<128, 128>(p, s);
  if (t64x128 >= 256) return launch_x3_nt<64, 128>(p, s);
  return launch_x3_nt<64, 64>(p, s);
}
