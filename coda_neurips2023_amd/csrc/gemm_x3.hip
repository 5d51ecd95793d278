// gemm_x3.hip -- fp32 GEMM on the bf16 matrix cores: every fp32 operand is carried as three bf16 pieces
// x = hi + mid + lo (8 + 8 + 8 significand bits: the fp32 value exactly; the residuals are computed without
// rounding) and a product a.b is the sum of ALL NINE piece products, each exact in fp32 (8 x 8 bits), accumulated
// in the fp32 MFMA accumulator.  Nothing of the operands is dropped: the result differs from an fp32 FMA chain only
// in the order and number of fp32 roundings of the accumulation (tests/test_gemm_x3_gpu.py measures both against
// float64).  Nine v_mfma_f32_32x32x16_bf16 cost 9/16 of the fp32 MFMAs they replace, and -- the point -- they run
// on the real matrix cores: the fp32 "MFMA" of this part shares the fp32 vector lanes (157 TFLOP/s either way,
// tools/mfma_valu_probe.hip), the bf16 one does not (2.5 PFLOP/s).  Splitting costs ~5 VALU instructions per
// operand ELEMENT, paid once per staged tile and amortised over the 64-128 outputs that element feeds -- unlike in
// the attention core, where the probabilities are produced on the fly (attention_bf16.hip, NS = 3).
//
//   C (M x N, row stride ldc) [+]= op(A) op(B) [+ bias],  op(A) M x K, op(B) K x N   (coda_gemm_f32's convention)
//
// Shape: 256 threads = 2 x 2 waves on a BM x BN tile of C (128 x 128, 64 x 128 or 64 x 64), K walked in steps of
// 32 through ONE LDS stage holding the three bf16 images of both tiles ([row][32 k], row stride 80 B: conflict-
// free ds_read_b128), the global loads of step i+1 in flight during the MFMAs of step i.  Either operand may be
// stored k-major (transa / !transb): its 4 x 4 blocks are then transposed on the way into LDS, so the MFMA side
// always reads 8 consecutive k of one output row / column.  Split-K (weight gradients: K = 16 384 token rows for
// 256 x 256 outputs): grid.z slices write partial tiles, summed in a fixed order by a second small kernel.
#include "coda_gemm.h"
#include "common.hip.h"

#include <cstdint>

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kThreadsX3 = 256;
constexpr int kBK = 32;          // k per stage
constexpr int kRowBytes = 80;    // 32 bf16 + 16 B pad

struct X3Params {
  const float *a, *b, *bias;
  float *c;
  long long lda, ldb, ldc;
  long long part_stride;  // floats between the partial outputs of consecutive k-slices (split-K), else 0
  int m, n, k;            // k: length of one k-slice
  int accumulate;
};

struct Pieces4 {
  bf16x4 p[3];
};
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

// One operand tile (ROWS output rows or columns x 32 k) on its way from global memory to the three LDS images.
// KMAJ = false: stored [row][k] (k contiguous): a thread owns float4 pieces of rows.
// KMAJ = true:  stored [k][row] (row contiguous): a thread owns 4 x 4 blocks and transposes them.
template <int ROWS, bool KMAJ>
struct Stage {
  static constexpr int PER = KMAJ ? (ROWS * kBK / 16 + kThreadsX3 - 1) / kThreadsX3 : ROWS * kBK / 4 / kThreadsX3;
  static constexpr int IMG = ROWS * kRowBytes;
  float4 v[KMAJ ? PER * 4 : PER];

  __device__ __forceinline__ void load(const float *g, long long ld, int row0, int k0, int tid) {
    if (!KMAJ) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int i = tid + u * kThreadsX3, row = i >> 3, c4 = i & 7;
        v[u] = *reinterpret_cast<const float4 *>(g + static_cast<size_t>(row0 + row) * ld + k0 + 4 * c4);
      }
    } else {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int blk = tid + u * kThreadsX3;
        if (ROWS * kBK / 16 < kThreadsX3 && blk >= ROWS * kBK / 16) continue;
        const int kb = blk / (ROWS / 4), rb = blk % (ROWS / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[4 * u + i] = *reinterpret_cast<const float4 *>(g + static_cast<size_t>(k0 + 4 * kb + i) * ld + row0 + 4 * rb);
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char *lds, int tid) const {
    if (!KMAJ) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int i = tid + u * kThreadsX3, row = i >> 3, c4 = i & 7;
        const Pieces4 s = split4(f32x4v{v[u].x, v[u].y, v[u].z, v[u].w});
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4 *>(lds + q * IMG + row * kRowBytes + 8 * c4) = s.p[q];
      }
    } else {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int blk = tid + u * kThreadsX3;
        if (ROWS * kBK / 16 < kThreadsX3 && blk >= ROWS * kBK / 16) continue;
        const int kb = blk / (ROWS / 4), rb = blk % (ROWS / 4);
        const float4 *w = &v[4 * u];
        const Pieces4 sx = split4(f32x4v{w[0].x, w[1].x, w[2].x, w[3].x});
        const Pieces4 sy = split4(f32x4v{w[0].y, w[1].y, w[2].y, w[3].y});
        const Pieces4 sz = split4(f32x4v{w[0].z, w[1].z, w[2].z, w[3].z});
        const Pieces4 sw = split4(f32x4v{w[0].w, w[1].w, w[2].w, w[3].w});
        unsigned char *base = lds + (4 * rb) * kRowBytes + 8 * kb;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          *reinterpret_cast<bf16x4 *>(base + q * IMG) = sx.p[q];
          *reinterpret_cast<bf16x4 *>(base + q * IMG + kRowBytes) = sy.p[q];
          *reinterpret_cast<bf16x4 *>(base + q * IMG + 2 * kRowBytes) = sz.p[q];
          *reinterpret_cast<bf16x4 *>(base + q * IMG + 3 * kRowBytes) = sw.p[q];
        }
      }
    }
  }
};

template <int BM, int BN, bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(kThreadsX3, 2) void gemm_x3_kernel(const X3Params p, int tiles_n) {
  constexpr int TM = BM / 64, TN = BN / 64;  // 32 x 32 MFMA tiles per wave in each direction
  constexpr int IMG_A = BM * kRowBytes, IMG_B = BN * kRowBytes;
  __shared__ __attribute__((aligned(16))) unsigned char s_a[3 * IMG_A];
  __shared__ __attribute__((aligned(16))) unsigned char s_b[3 * IMG_B];
  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const int tile = blockIdx.x, slice = blockIdx.z;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
  const int kbeg = slice * p.k;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stage<BM, A_KMAJ> sa;
  Stage<BN, B_KMAJ> sb;
  sa.load(p.a, p.lda, m0, kbeg, tid);
  sb.load(p.b, p.ldb, n0, kbeg, tid);
  for (int k0 = 0; k0 < p.k; k0 += kBK) {
    __syncthreads();  // everyone is done reading the previous stage
    sa.store(s_a, tid);
    sb.store(s_b, tid);
    __syncthreads();
    if (k0 + kBK < p.k) {  // next stage: in flight during this stage's MFMAs
      sa.load(p.a, p.lda, m0, kbeg + k0 + kBK, tid);
      sb.load(p.b, p.ldb, n0, kbeg + k0 + kBK, tid);
    }
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          fa[i][q] = *reinterpret_cast<const bf16x8 *>(s_a + q * IMG_A + (wm + 32 * i + l31) * kRowBytes + 32 * kk + 16 * half);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          fb[j][q] = *reinterpret_cast<const bf16x8 *>(s_b + q * IMG_B + (wn + 32 * j + l31) * kRowBytes + 32 * kk + 16 * half);
      // all nine piece products, smallest first; the (i, j) tiles interleave so that consecutive MFMAs are independent
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        constexpr int qa_of[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
        constexpr int qb_of[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
        const int qa = qa_of[t], qb = qb_of[t];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acc[i][j], 0, 0, 0);
      }
    }
  }
  // epilogue
  float *c = p.c + static_cast<size_t>(slice) * p.part_stride;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + l31;
      const float bias = (p.bias && slice == 0) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + crow_x3(r, half);
        float *dst = c + static_cast<size_t>(row) * p.ldc + col;
        float v = acc[i][j][r] + bias;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
}

// out (m x n, row stride ldc) [+]= sum over slices of partials (slices, m, n)
__global__ __launch_bounds__(256) void x3_reduce_kernel(const float *__restrict__ part, int slices, int m, int n,
                                                        float *__restrict__ out, long long ldc, int accumulate) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x, total = static_cast<size_t>(m) * n / 4;
  if (i >= total) return;
  const size_t e = i * 4, row = e / n, col = e % n;
  float4 s = *reinterpret_cast<const float4 *>(part + e);
  for (int z = 1; z < slices; ++z) {
    const float4 t = *reinterpret_cast<const float4 *>(part + static_cast<size_t>(z) * m * n + e);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  float4 *dst = reinterpret_cast<float4 *>(out + row * ldc + col);
  if (accumulate) {
    const float4 o = *dst;
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  *dst = s;
}

template <int BM, int BN>
int launch_x3(const X3Params &p, int transa, int transb, int slices, hipStream_t s) {
  const int tiles_n = p.n / BN;
  const dim3 grid(static_cast<unsigned>((p.m / BM) * tiles_n), 1, slices);
  if (!transa && transb) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, false, false>), grid, dim3(kThreadsX3), 0, s, p, tiles_n);
  else if (!transa && !transb) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, false, true>), grid, dim3(kThreadsX3), 0, s, p, tiles_n);
  else if (transa && !transb) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, true>), grid, dim3(kThreadsX3), 0, s, p, tiles_n);
  else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, false>), grid, dim3(kThreadsX3), 0, s, p, tiles_n);
  return launch_status();
}

}  // namespace
}  // namespace coda

CODA_API size_t coda_gemm_x3_workspace_bytes(int m, int n, int k) {
  // split-K partial tiles: at most 64 slices of m x n floats
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  return static_cast<size_t>(64) * m * n * sizeof(float);
}

CODA_API int coda_gemm_x3_f32(int transa, int transb, int m, int n, int k, const float *a, long long lda, const float *b,
                              long long ldb, float *c, long long ldc, const float *bias, int accumulate, void *workspace,
                              size_t workspace_bytes, void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (!a || !b || !c || k == 0) return CODA_EINVAL;
  // shapes / alignments this kernel takes (everything else: coda_gemm_f32)
  if (m % 64 || n % 64 || k % kBK || lda % 4 || ldb % 4 || ldc % 4 ||
      (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) % 16)
    return CODA_ENOSPC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  // tile choice: 128 x 128 when that still gives >= 2 workgroups per CU, else smaller tiles
  const long long t128 = (m % 128 == 0 && n % 128 == 0) ? static_cast<long long>(m / 128) * (n / 128) : 0;
  const long long t64x128 = (n % 128 == 0) ? static_cast<long long>(m / 64) * (n / 128) : 0;
  const long long t64 = static_cast<long long>(m / 64) * (n / 64);
  // split-K when even the smallest tiles leave most of the chip idle and K is long (weight gradients)
  int slices = 1;
  if (t64 < 256 && k >= 2048) {
    slices = static_cast<int>(min(64LL, max(1LL, 512 / t64)));
    while (slices > 1 && (k % (slices * kBK) != 0)) --slices;
  }
  X3Params p{a, b, bias, c, lda, ldb, ldc, 0, m, n, k / slices, accumulate};
  if (slices > 1) {
    const size_t need = static_cast<size_t>(slices) * m * n * sizeof(float);
    if (!workspace || workspace_bytes < need || ldc % 4) return CODA_ENOSPC;
    p.c = static_cast<float *>(workspace);
    p.ldc = n;
    p.part_stride = static_cast<long long>(m) * n;
    p.accumulate = 0;
    p.bias = nullptr;
    const int st = launch_x3<64, 64>(p, transa, transb, slices, s);
    if (st != CODA_OK) return st;
    const size_t quads = static_cast<size_t>(m) * n / 4;
    hipLaunchKernelGGL(x3_reduce_kernel, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, s,
                       static_cast<const float *>(workspace), slices, m, n, c, ldc, accumulate);
    if (bias) return CODA_EINVAL;  // no caller combines split-K with a bias
    return launch_status();
  }
  if (t128 >= 512) return launch_x3<128, 128>(p, transa, transb, 1, s);
  if (t64x128 >= 256) return launch_x3<64, 128>(p, transa, transb, 1, s);
  return launch_x3<64, 64>(p, transa, transb, 1, s);
}
