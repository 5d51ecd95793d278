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
// Shape: 128 x 128 (or 128 x 64) tiles of C, K walked in stages of 32 ([row][32 k] bf16 per plane in LDS, row stride
// 80 B: conflict-free ds_read_b128 fragments); see x3_nt_pipe_kernel for the pipeline and for the accumulator scheme
// that removes the matrix core's rounding bias.
#include "coda_gemm.h"
#include "common.hip.h"

#include <cstdint>
#include <cstdlib>

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
  const __bf16 *w;     // TILED planes (x3_split_kernel): block (row tile of 128, k stage of 32) = 3 planes x 128 x 32
  const float *bias;
  float *c;
  long long lda, ldc;
  int w_kt_total;      // k stages of the whole plane set (its column count / 32)
  int w_row0, w_kt0;   // first output column (a multiple of BN) and first k stage of this product inside the plane set
  int m, n, k;
  int accumulate;
  int tiles_m, tiles_n;
  int xcd_map;         // tiles of one A row block on one XCD
  int dbg;             // development probes (CODA_X3_DBG): parts of the pipelined kernel switched off
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

// ---- the kernel: persistent workgroups, producer and consumer waves -----------------------------------------------------
// A first form (256 threads, every wave loads, splits, stores and multiplies; two barriers per 32-k stage, two workgroups
// per CU) reached 38 % of the six-product MFMA pace: whatever a wave does besides MFMAs -- waiting for its loads,
// splitting the activations, writing LDS -- it does INSTEAD of issuing MFMAs.  Measured with the parts switched off one
// at a time, MFMAs + fragment reads alone, loads alone and split + LDS stores alone ADD UP to the kernel's time (a
// one-wave-per-SIMD variant with everything in one instruction stream: 102 us = 49 + 22 epilogue + 24 + 15 on
// 98 304 x 256 x 256), i.e. nothing overlapped.  So the roles are separated (8 waves, two per SIMD, one workgroup per CU):
//   waves 0-3 (consumers, a 64 x 64 quarter of the 128 x 128 tile each): fragment reads + 48 MFMAs per stage, the epilogue;
//   waves 4-7 (producers): global loads P stages ahead into a register ring, activation split, LDS stores.
// A workgroup walks a LIST of tiles as one flat sequence of stages, two LDS buffers, ONE barrier per stage:
//   stage s:  consumers on buffer s & 1  ||  producers: stage s + 1 registers -> buffer (s + 1) & 1, loads of stage s + 1 + P
// The two instruction streams of a SIMD interleave in hardware; a tile's epilogue runs under the producers' work for the
// next tile, and the first-load latency is paid once per workgroup instead of once per tile.
constexpr int kX3PipeThreads = 512;
template <int BN, int P>
__global__ __launch_bounds__(kX3PipeThreads, 2) void x3_nt_pipe_kernel(const X3NtParams p, int total_tiles, int nk) {
  constexpr int BM = 128;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int IMG_A = BM * kX3Row, IMG_W = BN * kX3Row;
  constexpr int BUF = 3 * IMG_A + 3 * IMG_W;
  constexpr int NA = BM * kX3BK / 4 / 256;         // 4 float4 pieces of A per producer thread and stage
  constexpr int NW = 3 * BN * 4 / 256;             // 6 | 3 16-byte pieces of the W planes
  constexpr int U = (P % 2 == 0) ? P : 2 * P;      // stages per trip of the unrolled loop: ring slot and LDS buffer static
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];

  const int lane = lane_id(), w = wave_id();
  const bool consumer = w < 4;                     // wave-uniform
  if (p.dbg & 8) {  // A/B: the consumers' instruction stream in front of the producers' at issue
    if (consumer) __builtin_amdgcn_s_setprio(1);
  }
  if (p.dbg & 16) {
    if (!consumer) __builtin_amdgcn_s_setprio(1);
  }
  const int nwg = static_cast<int>(gridDim.x), b = static_cast<int>(blockIdx.x);
  // this workgroup's tiles: with the XCD map, XCD x owns the row blocks tm = x (mod 8) and its nwg / 8 workgroups take
  // that list's tiles (n fastest) round-robin, so the tiles_n tiles of a row block run on one L2 at about the same time
  const int xcd = b & 7, slot = b >> 3, per_xcd = nwg >> 3, t8 = total_tiles >> 3;
  const int my_tiles = p.xcd_map ? (t8 > slot ? (t8 - slot + per_xcd - 1) / per_xcd : 0)
                                 : (total_tiles > b ? (total_tiles - b + nwg - 1) / nwg : 0);
  if (my_tiles == 0) return;
  auto coords = [&](int i, int &m0, int &n0) {
    int tm, tn;
    if (p.xcd_map) {
      const int g = slot + i * per_xcd;
      tm = (g / p.tiles_n) * 8 + xcd;
      tn = g % p.tiles_n;
    } else {
      const int t = b + i * nwg;
      tm = t / p.tiles_n;
      tn = t % p.tiles_n;
    }
    m0 = tm * BM;
    n0 = tn * BN;
  };
  const int total = my_tiles * nk;  // stages of this workgroup

  if (!consumer) {
    // ================================================= producers =================================================
    const int tid = static_cast<int>(threadIdx.x) - 256;
    // Which tile row a lane block stages: consecutive blocks of a store instruction's lane group (8 lanes per row for the
    // activation pieces, 4 for the weight pieces) take rows FOUR apart, not adjacent ones -- with 80-byte rows, rows r and
    // r + 1 overlap in four of the 32 write banks (PMC: a third of the LDS cycles of the kernel were bank conflicts, all
    // from these stores), rows r and r + 4 are 320 bytes = exactly half a bank period apart.
    auto spread = [](int t) { return (t & ~7) | ((t & 1) << 2) | ((t >> 1) & 3); };
    int a_row[NA], a_c4[NA], w_q[NW], w_row[NW], w_c[NW], w_goff[NW];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int i = tid + u * 256;
      a_row[u] = spread(i >> 3);
      a_c4[u] = i & 7;
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = tid + u * 256, rem = i % (BN * 4);
      w_q[u] = i / (BN * 4);
      w_row[u] = spread(rem >> 2);
      w_c[u] = rem & 3;
      w_goff[u] = w_q[u] * 4096 + w_row[u] * 32 + 8 * w_c[u];  // elements inside the 3 x 128 x 32 block
    }
    f32x4v ra[P][NA];
    u32x4v rw[P][NW];
    // SIGN PATTERN (see the consumers): the activation pieces of row r, 16-k block b go to LDS multiplied by
    // (-1)^(r + b).  Row and block parity are the same for all of a thread's pieces: one constant mask.
    const unsigned int flip = (((spread(tid >> 3) ^ ((tid & 7) >> 2)) & 1) != 0) ? 0x80000000u : 0u;
    // fetch stream: (tile f_i, k-step f_k); past the end it keeps re-reading the last stage (unconditional loads keep
    // the compiler's counted waits exact; the extra stages come from L2)
    int f_i = 0, f_k = 0, f_m0, f_n0;
    coords(0, f_m0, f_n0);
#define X3P_FETCH(SLOT)                                                                                               \
  {                                                                                                                   \
    _Pragma("unroll") for (int u = 0; u < NA; ++u)                                                                    \
      ra[SLOT][u] = *reinterpret_cast<const f32x4v *>(p.a + static_cast<size_t>(f_m0 + a_row[u]) * p.lda +            \
                                                      kX3BK * f_k + 4 * a_c4[u]);                                     \
    _Pragma("unroll") for (int u = 0; u < NW; ++u)                                                                    \
      rw[SLOT][u] = *reinterpret_cast<const u32x4v *>(                                                                \
          p.w + (static_cast<size_t>((p.w_row0 + f_n0) >> 7) * p.w_kt_total + p.w_kt0 + f_k) * (3 * 4096) +          \
          (((p.w_row0 + f_n0) & 64) << 5) + w_goff[u]);                                                               \
    if (f_k + 1 < nk) ++f_k;                                                                                          \
    else if (f_i + 1 < my_tiles) {                                                                                    \
      f_k = 0;                                                                                                        \
      ++f_i;                                                                                                          \
      coords(f_i, f_m0, f_n0);                                                                                        \
    }                                                                                                                 \
  }
#define X3P_STORE(SLOT, BUFI)                                                                                         \
  {                                                                                                                   \
    unsigned char *sa_ = x3_smem + (BUFI) * BUF, *sw_ = sa_ + 3 * IMG_A;                                              \
    _Pragma("unroll") for (int u = 0; u < NA; ++u) {                                                                  \
      const u32x4v bits_ = __builtin_bit_cast(u32x4v, ra[SLOT][u]) ^ flip;                                            \
      const Pieces4 sp = split4(__builtin_bit_cast(f32x4v, bits_));                                                   \
      _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                   \
        *reinterpret_cast<bf16x4 *>(sa_ + q * IMG_A + a_row[u] * kX3Row + 8 * a_c4[u]) = sp.p[q];                     \
    }                                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < NW; ++u)                                                                    \
      *reinterpret_cast<u32x4v *>(sw_ + w_q[u] * IMG_W + w_row[u] * kX3Row + 16 * w_c[u]) = rw[SLOT][u];              \
  }
    // prologue: P stages in flight, stage 0 into buffer 0
#pragma unroll
    for (int d = 0; d < P; ++d) X3P_FETCH(d);
    X3P_STORE(0, 0);
    X3P_FETCH(0);
    __syncthreads();
    for (int s = 0; s < total;) {
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (s >= total) break;
        // stage s + 1 (ring slot (s + 1) % P) -> the other buffer (its last readers passed the barrier at the end of
        // stage s - 1), then the freed register set takes stage s + 1 + P.  (Past the end the last stage is stored
        // again: harmless.)
        if (!(p.dbg & 1)) {
          if (!(p.dbg & 32)) X3P_STORE((uu + 1) % P, (uu & 1) ^ 1);
          if (!(p.dbg & 64)) X3P_FETCH((uu + 1) % P);
        }
        __syncthreads();
        ++s;
      }
    }
#undef X3P_FETCH
#undef X3P_STORE
    return;
  }
  // ================================================= consumers =================================================
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
  // THE MATRIX CORE'S ACCUMULATION IS BIASED: v_mfma_f32_32x32x16_bf16 truncates the aligned addends toward minus
  // infinity (measured, tools/x3_bias.py: mean signed error -2e-11 * K * mean|y| whatever the sign of y, where the fp32
  // MFMA and the library GEMM are unbiased).  Per element that is far below the rounding noise -- the RMS error of this
  // kernel is 0.6x the library's -- but it has ONE sign, so a sum over the 16 384 token rows of an output column (every
  // bias and LayerNorm gradient of the step is such a sum, and they cancel heavily) collects it 16 384-fold where
  // random errors collect 128-fold: whole-step gradients moved by 1.2e-3 against float64 (0.36e-3 with the library).
  // Cure: two accumulators, X for the even and Y for the odd 16-k blocks, with the activations' sign alternating over
  // rows AND blocks (the producers store (-1)^(row + block) * a), so that
  //   X[r] = s_r * (sum over even blocks),  Y[r] = -s_r * (sum over odd blocks),  y[r] = s_r * (X[r] - Y[r]),  s_r = (-1)^r:
  // both accumulators carry the same negative bias, the difference keeps only its row-to-row fluctuation, and what is
  // left changes sign from one row to the next -- column sums see noise again, not a drift.
  f32x16 acc[TM][TN], acs[TM][TN];  // X and Y
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acs[i][j][r] = 0.f;
  int c_i = 0, c_k = 0, c_m0, c_n0;
  coords(0, c_m0, c_n0);
  __syncthreads();  // (the producers' prologue barrier)
  for (int s = 0; s < total;) {
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      if (s >= total) break;
      const unsigned char *sa = x3_smem + uu * BUF, *sw = sa + 3 * IMG_A;
      if (!(p.dbg & 2))
#pragma unroll
      for (int kk = 0; kk < kX3BK / 16; ++kk) {
        bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            fa[i][q] = *reinterpret_cast<const bf16x8 *>(sa + q * IMG_A + (wm + 32 * i + l31) * kX3Row + 32 * kk + 16 * half);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            fb[j][q] = *reinterpret_cast<const bf16x8 *>(sw + q * IMG_W + (wn + 32 * j + l31) * kX3Row + 32 * kk + 16 * half);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          constexpr int qa_of[6] = {2, 0, 1, 1, 0, 0};
          constexpr int qb_of[6] = {0, 2, 1, 0, 1, 0};
          const int qa = qa_of[t], qb = qb_of[t];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (kk & 1) acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acs[i][j], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acc[i][j], 0, 0, 0);
            }
        }
      }
      if (c_k + 1 == nk && (p.dbg & 4)) {
        c_k = 0;
        ++c_i;
        if (c_i < my_tiles) coords(c_i, c_m0, c_n0);
      } else if (c_k + 1 == nk) {
        // tile done: plain stores (2 rows x 128 contiguous bytes per instruction), running under the producers' work
        // for the next tile
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = c_n0 + wn + 32 * j + l31;
            const float bias = p.bias ? p.bias[col] : 0.f;
            float *dst0 = p.c + static_cast<size_t>(c_m0 + wm + 32 * i + 4 * half) * p.ldc + col;
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)  // row = (r & 3) + 8 (r >> 2) + 4 half (+ even offsets): its parity is r & 1
              y[r] = (r & 1) ? acs[i][j][r] - acc[i][j][r] : acc[i][j][r] - acs[i][j][r];
            if (p.accumulate) {  // (hoisted: inside the element loop the compiler branched and waited per element)
              float old[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) old[r] = dst0[static_cast<size_t>(crow_x3(r, 0)) * p.ldc];
#pragma unroll
              for (int r = 0; r < 16; ++r) y[r] += old[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              dst0[static_cast<size_t>(crow_x3(r, 0)) * p.ldc] = y[r] + bias;
              acc[i][j][r] = acs[i][j][r] = 0.f;
            }
          }
        c_k = 0;
        ++c_i;
        if (c_i < my_tiles) coords(c_i, c_m0, c_n0);
      } else {
        ++c_k;
      }
      __syncthreads();
      ++s;
    }
  }
}

// ---- the weight gradients: dW = dY^T X over the token rows, as partial sums per token slice -------------------------------
// Both operands are activations (fp32, token-major: feature contiguous), the contraction runs over TOKENS.  The MFMA wants,
// per lane, eight consecutive k (tokens) of one output row / column (feature) -- the transpose of how the data lie.  They
// go to LDS as they are ([token][feature] bf16 per plane: the producers' stores stay whole-row and conflict-free) and the
// fragments come out through ds_read_b64_tr_b16, the transposing LDS read: within a 16-lane group, lane c receives
// element (c & 3) of the four 8-byte pieces addressed by lanes (c >> 2), 4 + (c >> 2), 8 + (c >> 2), 12 + (c >> 2)
// (tools/tr_probe.hip), i.e. with lane i pointing at token k0 + (i >> 2), features f0 + 4 (i & 3) .. + 3, lane c gets
// tokens k0 .. k0 + 3 of feature f0 + c: two such reads are one 32 x 16 MFMA operand.  Rows are 256 + 64 bytes apart so
// that the four token rows of a read fall on different banks.
//   part (slices, M, N) [slice s] = dY[s-th token range]^T X[same range],   dY (T x M), X (T x N), row strides lddy / ldx
// The sum over the slices is the caller's (the grouped column-sum kernel that already closes the library path's chunks).
// Same pipeline as x3_nt_pipe_kernel (producer / consumer waves, flat stage sequence over the workgroup's (slice, tile)
// list, X / Y accumulators with alternating signs); both operands are split by the producers.
struct X3TnParams {
  const float *dy, *x;
  float *part;
  long long lddy, ldx;
  int m, n;            // output rows (features of dY) and columns (features of X)
  int slice_rows;      // tokens per slice (a multiple of 32)
  int tiles_m, tiles_n, slices;
};
constexpr int kTnRow = 320;  // bytes per LDS token row: 128 features x 2 B + 64 B

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *plane, int token0, int feat0, int lane) {
  // the 32 (features feat0 ..) x 16 (tokens token0 ..) operand of v_mfma_f32_32x32x16_bf16 from a [token][feature] image
  const int g = lane >> 4, i = lane & 15;
  const unsigned char *a0 = plane + (token0 + 8 * (g >> 1) + (i >> 2)) * kTnRow + (feat0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0 + 4 * kTnRow));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int P>
__global__ __launch_bounds__(kX3PipeThreads, 2) void x3_tn_pipe_kernel(const X3TnParams p) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int IMG = kX3BK * kTnRow;              // one plane of one operand: 32 token rows
  constexpr int BUF = 6 * IMG;                     // dY planes, then X planes
  constexpr int NA = kX3BK * BM / 4 / 256;         // 4 float4 pieces per operand, producer thread and stage
  constexpr int U = (P % 2 == 0) ? P : 2 * P;
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];

  const int lane = lane_id(), w = wave_id();
  const bool consumer = w < 4;
  const int nwg = static_cast<int>(gridDim.x), b = static_cast<int>(blockIdx.x);
  const int tiles = p.tiles_m * p.tiles_n, total_tiles = tiles * p.slices;
  const int my_tiles = total_tiles > b ? (total_tiles - b + nwg - 1) / nwg : 0;
  if (my_tiles == 0) return;
  const int nk = p.slice_rows / kX3BK;
  // (slice, tile) of this workgroup's i-th item: consecutive workgroups share a token slice (its rows come from L2)
  auto coords = [&](int i, int &row0, int &m0, int &n0, int &slice) {
    const int t = b + i * nwg;
    slice = t / tiles;
    const int tt = t % tiles;
    row0 = slice * p.slice_rows;
    m0 = (tt / p.tiles_n) * BM;
    n0 = (tt % p.tiles_n) * BN;
  };
  const int total = my_tiles * nk;

  if (!consumer) {
    const int tid = static_cast<int>(threadIdx.x) - 256;
    int tok[NA], f4[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int i = tid + u * 256;
      tok[u] = i >> 5;      // 32 float4 per 128-feature row
      f4[u] = i & 31;
    }
    f32x4v ry[P][NA], rx[P][NA];
    int f_i = 0, f_k = 0, f_row0, f_m0, f_n0, f_slice;
    coords(0, f_row0, f_m0, f_n0, f_slice);
#define X3T_FETCH(SLOT)                                                                                               \
  {                                                                                                                   \
    _Pragma("unroll") for (int u = 0; u < NA; ++u) {                                                                  \
      const size_t r_ = static_cast<size_t>(f_row0 + kX3BK * f_k + tok[u]);                                           \
      ry[SLOT][u] = *reinterpret_cast<const f32x4v *>(p.dy + r_ * p.lddy + f_m0 + 4 * f4[u]);                         \
      rx[SLOT][u] = *reinterpret_cast<const f32x4v *>(p.x + r_ * p.ldx + f_n0 + 4 * f4[u]);                           \
    }                                                                                                                 \
    if (f_k + 1 < nk) ++f_k;                                                                                          \
    else if (f_i + 1 < my_tiles) {                                                                                    \
      f_k = 0;                                                                                                        \
      ++f_i;                                                                                                          \
      coords(f_i, f_row0, f_m0, f_n0, f_slice);                                                                       \
    }                                                                                                                 \
  }
    // sign pattern of the dY pieces: (-1)^(output row + 16-token block); a float4 holds output rows 4 f4 .. + 3
    // (parities 0 1 0 1), its token block is tok >> 4 = (u >> 1) for every thread
#define X3T_STORE(SLOT, BUFI)                                                                                         \
  {                                                                                                                   \
    unsigned char *sy_ = x3_smem + (BUFI) * BUF, *sx_ = sy_ + 3 * IMG;                                                \
    _Pragma("unroll") for (int u = 0; u < NA; ++u) {                                                                  \
      const u32x4v m_ = (u >> 1) ? u32x4v{0x80000000u, 0u, 0x80000000u, 0u} : u32x4v{0u, 0x80000000u, 0u, 0x80000000u}; \
      const Pieces4 py = split4(__builtin_bit_cast(f32x4v, __builtin_bit_cast(u32x4v, ry[SLOT][u]) ^ m_));            \
      const Pieces4 px = split4(rx[SLOT][u]);                                                                         \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                                 \
        *reinterpret_cast<bf16x4 *>(sy_ + q * IMG + tok[u] * kTnRow + 8 * f4[u]) = py.p[q];                           \
        *reinterpret_cast<bf16x4 *>(sx_ + q * IMG + tok[u] * kTnRow + 8 * f4[u]) = px.p[q];                           \
      }                                                                                                               \
    }                                                                                                                 \
  }
#pragma unroll
    for (int d = 0; d < P; ++d) X3T_FETCH(d);
    X3T_STORE(0, 0);
    X3T_FETCH(0);
    __syncthreads();
    for (int s = 0; s < total;) {
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (s >= total) break;
        X3T_STORE((uu + 1) % P, (uu & 1) ^ 1);
        X3T_FETCH((uu + 1) % P);
        __syncthreads();
        ++s;
      }
    }
#undef X3T_FETCH
#undef X3T_STORE
    return;
  }
  // consumers
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
  f32x16 acc[TM][TN], acs[TM][TN];  // X and Y (see x3_nt_pipe_kernel)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acs[i][j][r] = 0.f;
  int c_i = 0, c_k = 0, c_row0, c_m0, c_n0, c_slice;
  coords(0, c_row0, c_m0, c_n0, c_slice);
  __syncthreads();
  for (int s = 0; s < total;) {
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      if (s >= total) break;
      const unsigned char *sy = x3_smem + uu * BUF, *sx = sy + 3 * IMG;
#pragma unroll
      for (int kk = 0; kk < kX3BK / 16; ++kk) {
        bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[i][q] = tr_frag(sy + q * IMG, 16 * kk, wm + 32 * i, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[j][q] = tr_frag(sx + q * IMG, 16 * kk, wn + 32 * j, lane);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          constexpr int qa_of[6] = {2, 0, 1, 1, 0, 0};
          constexpr int qb_of[6] = {0, 2, 1, 0, 1, 0};
          const int qa = qa_of[t], qb = qb_of[t];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (kk & 1) acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acs[i][j], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fb[j][qb], acc[i][j], 0, 0, 0);
            }
        }
      }
      if (c_k + 1 == nk) {
        float *tile = p.part + (static_cast<size_t>(c_slice) * p.m + c_m0) * p.n + c_n0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float *dst0 = tile + static_cast<size_t>(wm + 32 * i + 4 * half) * p.n + wn + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              dst0[static_cast<size_t>(crow_x3(r, 0)) * p.n] =
                  (r & 1) ? acs[i][j][r] - acc[i][j][r] : acc[i][j][r] - acs[i][j][r];
              acc[i][j][r] = acs[i][j][r] = 0.f;
            }
          }
        c_k = 0;
        ++c_i;
        if (c_i < my_tiles) coords(c_i, c_row0, c_m0, c_n0, c_slice);
      } else {
        ++c_k;
      }
      __syncthreads();
      ++s;
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
  // TILED plane sets (what x3_nt_pipe_kernel stages): a matrix of R output columns x C contraction indices is stored as
  // blocks [R / 128][C / 32] of 3 planes x 128 rows x 32 k bf16 = 24 KB each, so that a stage's weight operand is ONE
  // contiguous run (whole cache lines, sequential 16-byte loads) instead of 128 x 3 row pieces of 64 bytes.
  // nt: R = rows, C = cols (needs cols % 32 == 0); nn: the transpose, R = cols, C = rows (needs rows % 32 == 0); rows past
  // R in the last row tile stay as the caller zeroed them.
  auto tiled = [](int r, int c, int kt_total, int plane) -> size_t {
    return ((static_cast<size_t>(r >> 7) * kt_total + (c >> 5)) * 3 + plane) * 4096 + (r & 127) * 32 + (c & 31);
  };
  if (t.nt[it]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + ty + 8 * q, c = c0 + tx;
      if (r < rows && c < cols) {
        __bf16 o[3];
        pieces(s[ty + 8 * q][tx], o);
#pragma unroll
        for (int k = 0; k < 3; ++k) t.nt[it][tiled(r, c, cols >> 5, k)] = o[k];
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
        for (int k = 0; k < 3; ++k) t.nn[it][tiled(c, r, rows >> 5, k)] = o[k];
      }
    }
  }
}

template <int BN>
int launch_x3_pipe(X3NtParams p, hipStream_t s) {
  constexpr size_t lds = 2 * (3 * 128 * kX3Row + 3 * BN * kX3Row);
  p.tiles_m = p.m / 128;
  p.tiles_n = p.n / BN;
  const int tiles = p.tiles_m * p.tiles_n;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    static int cached[64] = {0};
    if (dev >= 0 && dev < 64) {
      if (!cached[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cached[dev] = v;
        else cached[dev] = 256;
      }
      cus = cached[dev];
    }
  }
  int nwg = tiles < cus ? tiles : cus;
  p.xcd_map = (p.tiles_m % 8 == 0 && nwg % 8 == 0 && nwg >= 8) ? 1 : 0;
// (prefetch depths 2 / 3 / 4 measure the same, CODA_X3_DBG=8 -- the consumer waves at priority 1 -- 0 .. -4 %: round 6.
  // hipcc waits vmcnt(9) .. vmcnt(0) in front of a stage's LDS stores, i.e. it drains every load in flight and the
  // register ring is one stage deep whatever P; with the loads as inline-asm statements and a hand-counted
  // `s_waitcnt vmcnt((P - 1) * 10)` the ring really is P deep -- and the kernel exactly as fast: the producers are bound by
  // load THROUGHPUT (40 KB per stage and CU), not by latency.  Kept compiler-visible.)
  auto kern = x3_nt_pipe_kernel<BN, 3>;
  const int st = raise_dynamic_lds(kern, lds);
  if (st != CODA_OK) return st;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nwg)), dim3(kX3PipeThreads), lds, s, p, tiles, p.k / kX3BK);
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
      if ((it.nt && it.cols % 32) || (it.nn && it.rows % 32)) return CODA_EINVAL;  // the tiled layout's k stages
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

CODA_API int coda_gemm_x3_tn_f32(int rows, int m, int n, const float *dy, long long lddy, const float *x, long long ldx,
                                 float *part, int slices, void *stream) {
  using namespace coda;
  if (rows < 0 || m < 0 || n < 0 || slices < 1) return CODA_EINVAL;
  if (rows == 0 || m == 0 || n == 0) return CODA_OK;
  if (!dy || !x || !part) return CODA_EINVAL;
  if (m % 128 || n % 128 || rows % slices || (rows / slices) % kX3BK || lddy % 4 || ldx % 4 ||
      (reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) % 16 || reinterpret_cast<uintptr_t>(part) % 4)
    return CODA_ENOSPC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  X3TnParams p{dy, x, part, lddy, ldx, m, n, rows / slices, m / 128, n / 128, slices};
  constexpr int P = 2;
  constexpr size_t lds = 2 * 6 * kX3BK * kTnRow;
  const long long tiles = static_cast<long long>(p.tiles_m) * p.tiles_n * slices;
  const int nwg = static_cast<int>(tiles < 256 ? tiles : 256);
  auto kern = x3_tn_pipe_kernel<P>;
  const int st = raise_dynamic_lds(kern, lds);
  if (st != CODA_OK) return st;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nwg)), dim3(kX3PipeThreads), lds, s, p);
  return launch_status();
}

CODA_API int coda_gemm_x3_nt_f32(int m, int n, int k, const float *a, long long lda, const void *w_tiled, int w_cols,
                                 int w_row0, int w_col0, float *c, long long ldc, const float *bias, int accumulate,
                                 void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (!a || !w_tiled || !c || k == 0 || w_cols <= 0 || w_row0 < 0 || w_col0 < 0) return CODA_EINVAL;
  // shapes / alignments this kernel takes (everything else: coda_gemm_f32)
  const int bn = (n % 128 == 0 && w_row0 % 128 == 0) ? 128 : 64;
  if (m % 128 || n % 64 || k % kX3BK || lda % 4 || w_cols % kX3BK || w_col0 % kX3BK || w_col0 + k > w_cols || w_row0 % bn ||
      (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w_tiled)) % 16 ||
      reinterpret_cast<uintptr_t>(c) % 4)
    return CODA_ENOSPC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  static const int dbg = [] { const char *e = getenv("CODA_X3_DBG"); return e ? atoi(e) : 0; }();
  X3NtParams p{a, static_cast<const __bf16 *>(w_tiled), bias, c, lda, ldc, w_cols / kX3BK, w_row0, w_col0 / kX3BK,
               m, n, k, accumulate, 0, 0, 0, dbg};
  return bn == 128 ? launch_x3_pipe<128>(p, s) : launch_x3_pipe<64>(p, s);
}
