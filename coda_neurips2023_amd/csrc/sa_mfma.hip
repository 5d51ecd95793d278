// sa_mfma.hip -- the set-abstraction shared MLP as hand-written fp32-MFMA GEMMs (gfx950).
//
// Reference: SharedMLP([3, 64, 128, 256], bn=True) = per layer 1x1 Conv2d (no bias) -> BatchNorm2d -> ReLU, then
// F.max_pool2d over nsample (third_party_pointnet2/pointnet2/pytorch_utils.py:8-33, pointnet2_modules.py:247-253).
// Here the activations are channels-last rows (one row per distinct neighbour of a ball-query group, see
// include/coda_sa_mlp.h) and every 1x1 convolution is a GEMM on v_mfma_f32_32x32x2_f32 with the work around it
// fused in:
//
//   sa_fwd_kernel<CIN, COUT, FIRST, POOL>      y_out = relu(bn(y_in)) W^T
//       prologue  FIRST: y_in = x . w1^T recomputed from the grouped xyz (layer 1 never exists in memory)
//                 else : y_in read once; BN + ReLU of the layer below applied while the tile is staged into LDS
//       epilogue  per-channel sum / sum of squares (weighted by the row multiplicity) for this layer's batch
//                 statistics; POOL: running max of sign(gamma) * y per (group, channel) with its row index
//   sa_bwd_dx_kernel<CIN, COUT, LAST, FIRST>   dmid_in = relu'(.) (dy W)
//       prologue  dy = BN backward of the upstream gradient formed from y_out while the tile is staged; LAST: the
//                 upstream gradient is the pooled one (d, sel) -- the sparse tensor is never materialised
//       epilogue  ReLU mask of the layer below and that layer's BN-backward sums; FIRST: also sum dmid_in * x_j,
//                 from which layer 1's weight gradient follows in closed form (sa_l1_bwd_kernel)
//   sa_bwd_dw_kernel<CIN, COUT, LAST, FIRST>   dW = dy^T relu(bn(y_in)), K = the rows; per-workgroup partial tiles
//                                              in registers over the whole row range, fixed-order reduction
//
// Tiling.  A workgroup = 4 waves, one per SIMD (up to 512 VGPRs per lane), persistent over a contiguous range of
// 64-row sub-tiles.  fp32 MFMA runs at the vector rate (64 cycles per 32x32x2), so operand delivery is cheap: the
// stationary operand (the weights) lives in REGISTERS in MFMA fragment layout for the whole kernel, the moving
// operand goes through one LDS tile, the next sub-tile's global loads are in flight (registers) under the MFMAs
// of the current one.  MFMA operand maps (cdna_hip_programming.md section 3): A[i = lane & 31][k = lane >> 5],
// B[k = lane >> 5][j = lane & 31], D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#include "coda_sa_mlp.h"
#include "common.hip.h"

#include <cstdlib>
#include <mutex>

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 256;     // threads per workgroup: 4 waves, one per SIMD
constexpr int kRows = 64;   // rows of a sub-tile

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)): every barrier would wait for the global stores of the previous sub-tile and for the prefetch
// loads of the next one, which are meant to stay in flight under the MFMAs.  Global memory is not used for
// communication inside a workgroup here, so ordering LDS is all a barrier has to do.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ i32x4 ldg4i(const int32_t *p) { return *reinterpret_cast<const i32x4 *>(p); }

// y1[c] = x . w1[c] with one rounding per operation, the same expression as csrc/sa_mlp.hip's conv3()
__device__ __forceinline__ float dot3w(float x0, float x1, float x2, float w0, float w1, float w2) {
  return __fadd_rn(__fadd_rn(__fmul_rn(x0, w0), __fmul_rn(x1, w1)), __fmul_rn(x2, w2));
}
__device__ __forceinline__ float bn_act(float y, float scale, float shift) {
  return fmaxf(__fadd_rn(__fmul_rn(y, scale), shift), 0.0f);
}

// The workgroup's range of sub-tiles: the packed rows split evenly, in whole sub-tiles, over the grid.
struct Range {
  long long total;       // packed rows
  long long sub0, sub1;  // sub-tiles [sub0, sub1) of 64 rows
  long long per;         // sub-tiles per workgroup
};
__device__ __forceinline__ Range wg_range(const int32_t *__restrict__ goff, long long groups) {
  Range r;
  r.total = goff[groups];
  const long long nsub = (r.total + kRows - 1) / kRows;
  r.per = (nsub + gridDim.x - 1) / gridDim.x;
  r.sub0 = static_cast<long long>(blockIdx.x) * r.per;
  if (r.sub0 > nsub) r.sub0 = nsub;
  r.sub1 = r.sub0 + r.per < nsub ? r.sub0 + r.per : nsub;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// packing of the ball-query groups (no host read-back): count -> scan -> scatter (+ xyz moments)
// ---------------------------------------------------------------------------------------------------------------
// one wave per group: distinct rows = slot 0 + the slots whose index differs from slot 0's (ball_query pads with
// copies of the first hit, ball_query_gpu.cu:35-48; real hits are distinct ascending indices)
__global__ __launch_bounds__(kT) void pack_count_kernel(const int32_t *__restrict__ idx, int32_t *__restrict__ cnt,
                                                        long long groups, int s_len, int dedup) {
  const long long g = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (g >= groups) return;
  const int lane = threadIdx.x & 63;
  if (!dedup) {
    if (lane == 0) cnt[g] = s_len;
    return;
  }
  const int32_t *row = idx + g * s_len;
  const int32_t first = row[0];
  int n = 0;
  for (int j = lane; j < s_len; j += 64) n += (j == 0 || row[j] != first) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if (lane == 0) cnt[g] = n;
}

// one workgroup: exclusive scan of the counts -> group_offsets; zeroes the moments and the caller's accumulators
__global__ __launch_bounds__(1024) void pack_scan_kernel(const int32_t *__restrict__ cnt, int32_t *__restrict__ goff,
                                                         long long groups, double *__restrict__ moments,
                                                         double *__restrict__ zero, int nzero) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x;
  for (int i = t; i < 10; i += 1024) moments[i] = 0.0;
  for (int i = t; i < nzero; i += 1024) zero[i] = 0.0;
  const long long per = (groups + 1023) / 1024;
  const long long g0 = t * per, g1 = g0 + per < groups ? g0 + per : groups;
  int sum = 0;
  for (long long g = g0; g < g1; ++g) sum += cnt[g];
  s_part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = t >= o ? s_part[t - o] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int run = s_part[t] - sum;
  for (long long g = g0; g < g1; ++g) {
    goff[g] = run;
    run += cnt[g];
  }
  if (t == 1023) goff[groups] = s_part[1023];
}

// a thread per (group, slot), grid-stride: row group_offsets[g] + j <- grouped[g][j] for j < cnt[g]; the moments are
// reduced per workgroup first (10 double atomics per workgroup, a few thousand per call)
__global__ __launch_bounds__(kT) void pack_scatter_kernel(const float *__restrict__ grouped, const int32_t *__restrict__ cnt,
                                                          const int32_t *__restrict__ goff, float *__restrict__ x,
                                                          float *__restrict__ roww, int32_t *__restrict__ grow,
                                                          double *__restrict__ moments, long long groups, int s_len) {
  __shared__ float s_m[10][kT / 64];
  float m[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) m[k] = 0.0f;
  const long long slots = groups * s_len;
  for (long long i = static_cast<long long>(blockIdx.x) * kT + threadIdx.x; i < slots;
       i += static_cast<long long>(gridDim.x) * kT) {
    const long long g = i / s_len;
    const int j = static_cast<int>(i - g * s_len);
    const int c = cnt[g];
    if (j < c) {
      const long long r = static_cast<long long>(goff[g]) + j;
      const float *src = grouped + i * 3;
      const float x0 = src[0], x1 = src[1], x2 = src[2];
      const float w = j == 0 ? static_cast<float>(s_len - c + 1) : 1.0f;
      x[r * 3] = x0; x[r * 3 + 1] = x1; x[r * 3 + 2] = x2;
      roww[r] = w;
      grow[r] = static_cast<int32_t>((g << 6) | j);  // group and row-in-group (s_len <= 64) in one word
      m[0] += w; m[1] += w * x0; m[2] += w * x1; m[3] += w * x2;
      m[4] += w * x0 * x0; m[5] += w * x0 * x1; m[6] += w * x0 * x2;
      m[7] += w * x1 * x1; m[8] += w * x1 * x2; m[9] += w * x2 * x2;
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    float v = m[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s_m[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    double v = 0.0;
    for (int q = 0; q < kT / 64; ++q) v += static_cast<double>(s_m[threadIdx.x][q]);
    atomicAdd(moments + threadIdx.x, v);
  }
}

// layer 1's batch statistics from the moments: y1 = x . w1[c] is linear in x
__global__ void l1_sums_kernel(const double *__restrict__ mom, const float *__restrict__ w1, double *__restrict__ sums, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const double a = w1[3 * i], b = w1[3 * i + 1], d = w1[3 * i + 2];
  sums[i] = a * mom[1] + b * mom[2] + d * mom[3];
  sums[c + i] = a * a * mom[4] + b * b * mom[7] + d * d * mom[9] + 2.0 * (a * b * mom[5] + a * d * mom[6] + b * d * mom[8]);
}

// layer 1 backward in closed form (include/coda_sa_mlp.h): with dy1 = a (d1 - w (m1 + xhat m2)),
//   dW1[c][j] = sum_p dy1 x_j = a (T_j - m1 X_j - m2 invstd ((W1[c] . XX[:, j]) - mean X_j))
__global__ void l1_bwd_kernel(const double *__restrict__ s5, const double *__restrict__ sbn, double n,
                              const float *__restrict__ gamma, const float *__restrict__ stats,
                              const double *__restrict__ mom, const float *__restrict__ w1, float *__restrict__ dw1,
                              float *__restrict__ dbeta, float *__restrict__ dgamma, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  if (dbeta) dbeta[i] = static_cast<float>(s5[i]);
  if (dgamma) dgamma[i] = static_cast<float>(s5[c + i]);
  if (!dw1) return;
  const double mean = stats[2 * c + i], invstd = stats[3 * c + i];
  const double a = static_cast<double>(__fmul_rn(gamma[i], stats[3 * c + i]));
  const double m1 = n > 0.0 ? sbn[i] / n : 0.0, m2 = n > 0.0 ? sbn[c + i] / n : 0.0;
  const double wa = w1[3 * i], wb = w1[3 * i + 1], wc = w1[3 * i + 2];
  const double xx[3][3] = {{mom[4], mom[5], mom[6]}, {mom[5], mom[7], mom[8]}, {mom[6], mom[8], mom[9]}};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double wxx = wa * xx[0][j] + wb * xx[1][j] + wc * xx[2][j];
    const double t = s5[(2 + j) * c + i];
    dw1[3 * i + j] = static_cast<float>(a * (t - m1 * mom[1 + j] - m2 * invstd * (wxx - mean * mom[1 + j])));
  }
}

// -DCODA_SA_PROF (tools/sa_prof.py builds a private copy of the library with it): shader-clock sums of the phases of a
// sub-tile per (kernel, workgroup < 64, wave), read back with coda_sa_prof_read.  Compiles to nothing in the library.
#ifdef CODA_SA_PROF
__device__ unsigned long long g_sa_prof[6][64][4][16];
#define SA_PROF_DECL unsigned long long prof_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t_ = __builtin_readcyclecounter()
#define SA_PROF_MARK(i)                                               \
  do {                                                                \
    __builtin_amdgcn_sched_barrier(0);                                \
    const unsigned long long now_ = __builtin_readcyclecounter();     \
    __builtin_amdgcn_sched_barrier(0);                                \
    prof_[i] += now_ - prof_t_;                                       \
    prof_t_ = now_;                                                   \
  } while (0)
#define SA_PROF_COUNT(i, v) prof_[i] += (v)
#define SA_PROF_STORE(kind)                                                                              \
  do {                                                                                                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64)                                                      \
      for (int q_ = 0; q_ < 16; ++q_) g_sa_prof[kind][blockIdx.x][threadIdx.x >> 6][q_] = prof_[q_];     \
  } while (0)
#else
#define SA_PROF_DECL
#define SA_PROF_MARK(i)
#define SA_PROF_COUNT(i, v)
#define SA_PROF_STORE(kind)
#endif

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
struct FwdArgs {
  const float *src, *w1, *st_in, *w, *roww;
  const int32_t *goff, *grow;
  long long groups;
  float *y_out;
  double *sums;
  const float *gamma;
  float *ysel;
  int32_t *sel;
  float *part_y;
  int32_t *part_sel, *part_gid;
};

// One workgroup per CU (up to 512 registers per lane: the weight fragments alone take 128).  A build for two workgroups
// per CU -- the input and output LDS tiles sharing their memory, no register prefetch -- would let one workgroup's MFMAs
// run under the other's staging / epilogue, but needs <= 256 registers: measured with 332 B of scratch per lane, layer 3
// took 0.46 ms instead of 0.33.  (PMC, profiles/r04_pmc_sa_mlp.md: 39 % MFMA-busy, 21 % of the wave cycles parked.)
constexpr bool fwd_alias(int cin, int cout) { return sizeof(float) * kRows * (cin + cout + 8) > 80 * 1024; }

template <int CIN, int COUT, bool FIRST, bool POOL>
__global__ __launch_bounds__(kT) void sa_fwd_kernel(const FwdArgs a) {
  static_assert(COUT % 128 == 0 && CIN % 32 == 0, "tile shape");
  static_assert(!POOL || COUT == kT, "pooling scan: one thread per channel");
  constexpr int CBW = COUT / 128;    // 32-column blocks per wave
  constexpr int KK = CIN / 2;        // MFMA k-steps
  constexpr int SA = CIN + 4;        // LDS row stride of the A tile (floats): b128 fragment reads conflict-free
  constexpr int SO = COUT + 4;       // LDS row stride of the output tile [64][COUT]
  constexpr int QPR = CIN / 4;       // float4 per input row
  constexpr int RPP = kT / QPR;      // rows per staging pass
  constexpr int NPASS = kRows / RPP;
  // Wide layers: the input and the output tile SHARE their memory (one more barrier per sub-tile), so that two
  // workgroups fit a CU's 160 KB and one's MFMAs run under the other's staging / epilogue / pooling scan.
  constexpr bool ALIAS = fwd_alias(CIN, COUT);
  extern __shared__ float lds[];
  float *s_a = lds;                                  // [64][SA], column k at (k & 1) * CIN/2 + (k >> 1)
  float *s_o = ALIAS ? lds : lds + kRows * SA;       // [64][SO] the sub-tile's output: pooled from here and written to
                                                     // memory as whole rows (16-byte stores)
  float *s_w = lds + (ALIAS ? kRows * SO : kRows * (SA + SO));  // [64] row multiplicity (0: not valid)
  int *s_grow = reinterpret_cast<int *>(s_w + kRows);  // [64] (group << 6) | row-in-group
  float *s_w1 = reinterpret_cast<float *>(s_grow + kRows);  // FIRST: [CIN][4] = w1[k][0..2], 0 ; then [CIN][2] scale, shift

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const Range rg = wg_range(a.goff, a.groups);
  const int b = blockIdx.x;
  if (POOL && tid == 0) a.part_gid[b] = -1;
  if (rg.sub0 >= rg.sub1) return;

  // stationary operand: W^T fragments, B[k = 2 kk + h][j = column] = w[column][2 kk + h]
  const int cbase = wv * (COUT / 4);
  float wreg[CBW][KK];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) wreg[cb][kk] = a.w[static_cast<size_t>(cbase + 32 * cb + l31) * CIN + 2 * kk + h];

  if (FIRST) {
    for (int k = tid; k < CIN; k += kT) {
      s_w1[4 * k] = a.w1[3 * k]; s_w1[4 * k + 1] = a.w1[3 * k + 1]; s_w1[4 * k + 2] = a.w1[3 * k + 2]; s_w1[4 * k + 3] = 0.f;
      s_w1[4 * CIN + 2 * k] = a.st_in[k]; s_w1[4 * CIN + 2 * k + 1] = a.st_in[CIN + k];
    }
  }
  // staging roles
  const int cq = tid % QPR, r0 = tid / QPR;       // non-FIRST: this thread's channel quad and first row
  f32x4 sc4 = {0, 0, 0, 0}, sh4 = {0, 0, 0, 0};
  if (!FIRST) { sc4 = ldg4(a.st_in + 4 * cq); sh4 = ldg4(a.st_in + CIN + 4 * cq); }
  const int frow = tid >> 2, fpart = tid & 3;     // FIRST: row and quarter of the channels

  f32x4 pre[FIRST ? 1 : NPASS];
  float px0 = 0.f, px1 = 0.f, px2 = 0.f;
  float pw = 0.f;
  int pg = 0;
  auto prefetch = [&](long long sub) {
    const long long s0 = sub * kRows;
    if (s0 + kRows <= rg.total) {  // a whole sub-tile (all but the last one): uniform base + constant per-thread offsets,
                                   // no bounds test per load (see DyStage::load_row)
      if (FIRST) {
        const float *xb = a.src + s0 * 3;
        px0 = xb[frow * 3]; px1 = xb[frow * 3 + 1]; px2 = xb[frow * 3 + 2];
      } else {
        const float *sb = a.src + s0 * CIN;
#pragma unroll
        for (int j = 0; j < NPASS; ++j) pre[j] = ldg4(sb + (r0 + RPP * j) * CIN + 4 * cq);
      }
      if (tid < kRows) {
        pw = (a.roww + s0)[tid];
        pg = (a.grow + s0)[tid];
      }
      return;
    }
    if (FIRST) {
      const long long r = s0 + frow;
      const bool ok = r < rg.total;
      px0 = ok ? a.src[r * 3] : 0.f; px1 = ok ? a.src[r * 3 + 1] : 0.f; px2 = ok ? a.src[r * 3 + 2] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < NPASS; ++j) {
        const long long r = s0 + r0 + RPP * j;
        pre[j] = r < rg.total ? ldg4(a.src + r * CIN + 4 * cq) : f32x4{0, 0, 0, 0};
      }
    }
    if (tid < kRows) {
      const long long r = s0 + tid;
      pw = r < rg.total ? a.roww[r] : 0.f;
      pg = r < rg.total ? a.grow[r] : -1;
    }
  };

  float st_s[CBW][2] = {}, st_q[CBW][2] = {};  // per-lane partial statistics of its column(s)
  // pooling state of this thread's channel
  const float sg = POOL ? (a.gamma[tid % COUT] >= 0.0f ? 1.0f : -1.0f) : 1.0f;
  int cur_g = -1, arg = 0;
  bool tail = false;  // the current group began in the previous workgroup
  float best = -INFINITY;
  const long long row_first = rg.sub0 * kRows;
  // (scalar group id -> the row base is a scalar address; the thread adds its channel)
  auto flush = [&]() {
    const float yv = best * sg;
    if (tail) {  // this workgroup holds the group's tail only
      (a.part_y + static_cast<size_t>(b) * COUT)[tid] = yv;
      (a.part_sel + static_cast<size_t>(b) * COUT)[tid] = arg;
      if (tid == 0) a.part_gid[b] = cur_g;
    } else {
      (a.ysel + static_cast<size_t>(cur_g) * COUT)[tid] = yv;
      (a.sel + static_cast<size_t>(cur_g) * COUT)[tid] = arg;
    }
  };

  // the same loads for a WHOLE sub-tile, in pieces: piece i is issued from step i of the matrix loop (constant i)
  constexpr int kParts = (FIRST ? 1 : NPASS) + 1;
  static_assert(kParts <= KK / 4, "one piece per MFMA step");
  auto prefetch_part = [&](long long sub, int i) {
    const long long s0 = sub * kRows;
    if (i < kParts - 1) {
      if (FIRST) {
        const float *xb = a.src + s0 * 3;
        px0 = xb[frow * 3]; px1 = xb[frow * 3 + 1]; px2 = xb[frow * 3 + 2];
      } else {
        const float *sb = a.src + s0 * CIN;
        pre[i] = ldg4(sb + (r0 + RPP * i) * CIN + 4 * cq);
      }
    } else if (tid < kRows) {
      pw = (a.roww + s0)[tid];
      pg = (a.grow + s0)[tid];
    }
  };
  prefetch(rg.sub0);
  SA_PROF_DECL;
  for (long long sub = rg.sub0; sub < rg.sub1; ++sub) {
    const long long s0 = sub * kRows;
    lds_barrier();  // the previous sub-tile's fragment reads / pooling scan are done
    SA_PROF_MARK(0);
    SA_PROF_COUNT(9, 1);
    if (FIRST) {
      // rows past the end: their outputs are zeroed by a MULTIPLICATION with 0 / 1.  Written as `ok ? f(x) : 0` the
      // compiler put every channel's constant reads and arithmetic under an exec-mask branch of its own: 16 serialised
      // LDS round trips per thread and sub-tile (tools/sa_prof.py: 5.2 k of the narrow forward's 13.5 k clocks)
      const float okf = s0 + frow < rg.total ? 1.0f : 0.0f;
#pragma unroll
      for (int i = 0; i < CIN / 4; i += 4) {
        // channels 16 j + 4 fpart + u: the four lanes of a row write ADJACENT 8-byte pieces, so a wave's 64 b64 writes
        // touch every bank exactly twice (with a contiguous quarter of the channels per lane -- 8 floats apart -- the
        // lanes of neighbouring rows collided: 44 % of the kernel's LDS cycles were bank conflicts)
        const int k0 = 4 * i + 4 * fpart;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 wk = *reinterpret_cast<const f32x4 *>(s_w1 + 4 * (k0 + u));
          const f32x2 ss = *reinterpret_cast<const f32x2 *>(s_w1 + 4 * CIN + 2 * (k0 + u));
          v[u] = okf * bn_act(dot3w(px0, px1, px2, wk[0], wk[1], wk[2]), ss[0], ss[1]);
        }
        *reinterpret_cast<f32x2 *>(s_a + frow * SA + (k0 >> 1)) = f32x2{v[0], v[2]};
        *reinterpret_cast<f32x2 *>(s_a + frow * SA + CIN / 2 + (k0 >> 1)) = f32x2{v[1], v[3]};
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPASS; ++j) {
        const int row = r0 + RPP * j;
        const bool ok = s0 + row < rg.total;
        const f32x4 y = pre[j];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ok ? bn_act(y[u], sc4[u], sh4[u]) : 0.f;
        *reinterpret_cast<f32x2 *>(s_a + row * SA + 2 * cq) = f32x2{v[0], v[2]};
        *reinterpret_cast<f32x2 *>(s_a + row * SA + CIN / 2 + 2 * cq) = f32x2{v[1], v[3]};
      }
    }
    if (tid < kRows) { s_w[tid] = pw; s_grow[tid] = pg; }
    SA_PROF_MARK(1);
    lds_barrier();
    SA_PROF_MARK(2);
    const bool more = sub + 1 < rg.sub1;
    const bool whole = more && (sub + 2) * kRows <= rg.total;  // the next sub-tile is a whole one: loads ride the loop
    if (__builtin_expect(more && !whole, 0)) prefetch(sub + 1);

    f32x16 acc[2][CBW];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rb][cb][e] = 0.f;
#pragma unroll
    for (int kq = 0; kq < KK / 4; ++kq) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(s_a + l31 * SA + h * (CIN / 2) + 4 * kq);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(s_a + (32 + l31) * SA + h * (CIN / 2) + 4 * kq);
      if (kq < kParts && __builtin_expect(whole, 1)) prefetch_part(sub + 1, kq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
          acc[0][cb] = mfma(a0[i], wreg[cb][4 * kq + i], acc[0][cb]);
          acc[1][cb] = mfma(a1[i], wreg[cb][4 * kq + i], acc[1][cb]);
        }
    }

    // ---- epilogue: statistics, store, pooling tile
    SA_PROF_MARK(3);
    if (ALIAS) lds_barrier();  // every wave's fragment reads of s_a are done
    SA_PROF_MARK(4);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rowq = 32 * rb + 8 * q + 4 * h;  // rows rowq .. rowq + 3 <-> registers 4 q .. 4 q + 3
        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(s_w + rowq);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
          const int col = cbase + 32 * cb + l31;
          f32x4 y4;
#pragma unroll
          for (int u = 0; u < 4; ++u) y4[u] = acc[rb][cb][4 * q + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float wy = w4[u] * y4[u];
            st_s[cb][u & 1] += wy;
            st_q[cb][u & 1] = fmaf(wy, y4[u], st_q[cb][u & 1]);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) s_o[(rowq + u) * SO + col] = y4[u];  // 32 consecutive floats per half-wave
        }
      }
    }
    SA_PROF_MARK(5);
    lds_barrier();
    SA_PROF_MARK(6);
    const long long left = rg.total - s0;
    const int nvalid = left < kRows ? static_cast<int>(left) : kRows;
    if (POOL) {  // thread = channel: running max of sign(gamma) * y over the rows of a group, lowest row on ties
      // The group structure of the sub-tile is the same for every channel: lane r of each wave looks at row r once
      // (does it start a group?), the ballot is a scalar bit mask, and the scan itself is three vector instructions
      // per row behind a scalar bit test -- no per-row LDS round trip for the group word, no per-row validity test
      // (rows past the end carry group -1, which is never flushed).  The values come 16 rows at a time.
      const int mygrow = s_grow[lane];
      const int prevg = lane ? (s_grow[lane - 1] >> 6) : cur_g;
      const unsigned long long chg = __ballot((mygrow >> 6) != prevg);
      int base = 0;  // row-in-group of row r = base + r while the group lasts
      if (cur_g >= 0 && (chg & 1ull) == 0ull) base = (__builtin_amdgcn_readlane(mygrow, 0) & 63);
      // The channel's 64 values in one LDS round trip (the accumulators are dead here: registers are free), then four
      // rows per scalar test: three of four quads hold no group start (3-4 groups per sub-tile).  Measured
      // (tools/sa_prof.py, clocks per sub-tile): row at a time with its own LDS reads 13.6 k, this form 7.1 k -- 6.3 k
      // with the no-start quads marked likely (the compiler had put THEM out of line) --, the same as a rolled loop over
      // quads 8.9 k; the pooling as a separate pass over the stored rows (328 MB re-read) was 33 us slower than this
      // scan in the step.  Of the 6.3 k: the compare chain 2.4 k (tools/scan_probe.hip), the 3-4 group ends 1.5 k (cold
      // code: each quad has its own copy), the 64 LDS reads 0.5 k.
      float vall[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) vall[r] = s_o[r * SO + tid];
#pragma unroll
      for (int rq = 0; rq < kRows; rq += 4) {
        if (__builtin_expect(((chg >> rq) & 0xfull) == 0ull, 1)) {  // (likely: the quads WITH a start go out of line)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float v = vall[rq + u] * sg;
            if (v > best) { best = v; arg = base + rq + u; }
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int r = rq + u;
            if ((chg >> r) & 1ull) {  // a scalar branch
              if (cur_g >= 0) flush();
              const int packed = __builtin_amdgcn_readlane(mygrow, r);
              const int rin = packed & 63;
              cur_g = packed >> 6;
              tail = s0 + r - rin < row_first;
              best = -INFINITY;
              arg = rin;
              base = rin - r;
            }
            const float v = vall[r] * sg;
            if (v > best) { best = v; arg = base + r; }
          }
        }
      }
    }
    SA_PROF_MARK(8);
    if (a.y_out) {  // whole rows, 16 bytes per lane
      constexpr int QO = COUT / 4, RO = kT / QO;
      const int oq = tid % QO, or0 = tid / QO;
      if (__builtin_expect(nvalid == kRows, 1)) {  // (uniform) no per-row test: the tile is read from LDS in one go, then stored
        float *ob = a.y_out + s0 * COUT;
        f32x4 t[kRows / RO];
#pragma unroll
        for (int j = 0; j < kRows / RO; ++j) t[j] = *reinterpret_cast<const f32x4 *>(s_o + (or0 + RO * j) * SO + 4 * oq);
#pragma unroll
        for (int j = 0; j < kRows / RO; ++j) *reinterpret_cast<f32x4 *>(ob + (or0 + RO * j) * COUT + 4 * oq) = t[j];
      } else {
#pragma unroll
        for (int j = 0; j < kRows / RO; ++j) {
          const int row = or0 + RO * j;
          if (row < nvalid)
            *reinterpret_cast<f32x4 *>(a.y_out + (s0 + row) * COUT + 4 * oq) = *reinterpret_cast<const f32x4 *>(s_o + row * SO + 4 * oq);
        }
      }
    }
    SA_PROF_MARK(7);
  }
  if (POOL && cur_g >= 0) flush();
  SA_PROF_STORE(POOL ? 0 : 1);

  // statistics: the two half-waves hold the two row halves of a column
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    float s = st_s[cb][0] + st_s[cb][1], q = st_q[cb][0] + st_q[cb][1];
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (h == 0) {
      const int col = cbase + 32 * cb + l31;
      atomicAdd(a.sums + col, static_cast<double>(s));
      atomicAdd(a.sums + COUT + col, static_cast<double>(q));
    }
  }
}

// merge the partial pools of the groups that straddle two workgroups, then out = relu(bn(ysel))
__global__ __launch_bounds__(kT) void pool_finish_kernel(float *__restrict__ ysel, int32_t *__restrict__ sel,
                                                         const float *__restrict__ part_y, const int32_t *__restrict__ part_sel,
                                                         const int32_t *__restrict__ part_gid, const int32_t *__restrict__ goff,
                                                         const float *__restrict__ gamma, const float *__restrict__ stats,
                                                         float *__restrict__ out, long long groups, int c, int nblk) {
  const long long total = goff[groups];
  const long long nsub = (total + kRows - 1) / kRows;
  const long long per = (nsub + nblk - 1) / nblk;
  for (long long i = blockIdx.x * static_cast<long long>(kT) + threadIdx.x; i < groups * c;
       i += static_cast<long long>(gridDim.x) * kT) {
    const long long g = i / c;
    const int ch = static_cast<int>(i - g * c);
    float y = ysel[i];
    int s = sel[i];
    const long long first = goff[g], last = static_cast<long long>(goff[g + 1]) - 1;
    const long long b0 = (first / kRows) / per, b1 = (last / kRows) / per;
    if (b1 != b0 && part_gid[b1] == static_cast<int32_t>(g)) {
      const float sg = gamma[ch] >= 0.0f ? 1.0f : -1.0f;
      const float y2 = part_y[b1 * c + ch];
      if (y2 * sg > y * sg) {  // strict: on ties the head part (lower rows) wins
        y = y2;
        s = part_sel[b1 * c + ch];
      }
      ysel[i] = y;
      sel[i] = s;
    }
    out[i] = fmaxf(__fadd_rn(__fmul_rn(y, stats[ch]), stats[c + ch]), 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
struct BwdArgs {
  const float *y_out, *dmid, *d;
  const int32_t *sel;
  const float *ca, *cm1, *cm2, *cmean, *cinv;  // per output channel: a = gamma invstd, m1, m2, mean, invstd
  const float *w;                              // (COUT, CIN)
  const float *src_in, *w1, *st_in;            // input activations: y_in (rows, CIN) or x (rows, 3) + w1
  const float *roww;
  const int32_t *goff, *grow;
  long long groups;
  float *dmid_in;
  double *sums_in;
  float *partials;
};

// dy of one float4 of a row.  dy = a (dsel - w (m1 + (y - mean) invstd m2)) evaluated as  a dsel + w (A + B y)  with the
// per-channel constants A = -a (m1 - mean invstd m2), B = -a invstd m2: two fused multiply-adds per element instead
// of seven operations (the staging of a dy tile is VALU work that the fp32 MFMAs do not overlap).  The rearrangement
// costs eps |mean| / std of relative accuracy (the two terms of A + B y cancel to the size of the centred value),
// ~1e-6 here against the 1e-3 bar.
__device__ __forceinline__ f32x4 dy4(f32x4 y, f32x4 adsel, float w, f32x4 ca, f32x4 cb) {
  f32x4 o;
#pragma unroll
  for (int u = 0; u < 4; ++u) o[u] = fmaf(w, fmaf(cb[u], y[u], ca[u]), adsel[u]);
  return o;
}

// Staging of the dy tile, shared by the dx and the dw kernel.  Thread = (channel quad cq, rows r0 NPASS + j): a
// thread's NPASS rows are CONSECUTIVE, so that they belong to one or two groups almost always (a sub-tile of 64 rows
// holds 3-4 groups on average: with rows dealt round-robin a third of the rows met a "third" group).
// LAST: the upstream gradient is the pooled one; a wave owns whole rows (COUT == 256), so the group of a row is
// wave-uniform and every test on it is a scalar branch.  The (d, sel) quads of the groups of the wave's first and last
// row of the NEXT sub-tile are prefetched with the tile; a row of a third group in between reads them directly.
template <int COUT, bool LAST>
struct DyStage {
  static constexpr int QPR = COUT / 4;
  static constexpr int RPP = kT / QPR;
  static constexpr int NPASS = kRows / RPP;
  f32x4 py[NPASS];
  f32x4 pd[LAST ? 1 : NPASS];
  f32x4 ca, cA, cB;  // a, A, B of this thread's channel quad (dy4)
  int cq, r0;
  // LAST: prefetched (d, sel) of two groups; nA / nB: their ids one sub-tile further ahead (so that the (d, sel)
  // loads of a prefetch never wait for an id load issued in the same prefetch), -2 = not loaded yet
  int nA = -2, nB = -2;
  int gA = -1, gB = -1;
  f32x4 dA = {0, 0, 0, 0}, dB = {0, 0, 0, 0};
  i32x4 sA = {0, 0, 0, 0}, sB = {0, 0, 0, 0};

  __device__ __forceinline__ int row_of(int j) const { return r0 * NPASS + j; }
  __device__ __forceinline__ void init(const BwdArgs &a) {
    cq = threadIdx.x % QPR;
    r0 = threadIdx.x / QPR;
    ca = ldg4(a.ca + 4 * cq);
    const f32x4 m1 = ldg4(a.cm1 + 4 * cq), m2 = ldg4(a.cm2 + 4 * cq), mu = ldg4(a.cmean + 4 * cq), is = ldg4(a.cinv + 4 * cq);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cA[u] = -ca[u] * (m1[u] - mu[u] * is[u] * m2[u]);
      cB[u] = -ca[u] * is[u] * m2[u];
    }
  }
  // FULL: every row of the sub-tile exists -- no per-load bounds test, and the addresses are a wave-uniform base (the
  // sub-tile's first row: scalar registers) plus an offset that never changes for this thread.  The loads of a whole
  // sub-tile are issued ONE PER MFMA STEP from inside the matrix loop of the sub-tile before (load_row / load_groups
  // called with constant j): in front of the loop, the ~30 loads of a wave -- each an exec-mask branch in the general
  // form -- were issued with the matrix pipe idle (tools/sa_prof.py: 4-6 thousand clocks per sub-tile).
  template <bool FULL>
  __device__ __forceinline__ void load_row(const BwdArgs &a, long long s0, long long total, int j) {
    const float *yb = a.y_out + s0 * COUT;
    const float *db = LAST ? nullptr : a.dmid + s0 * COUT;
    const int off = row_of(j) * COUT + 4 * cq;
    if (FULL) {
      py[j] = ldg4(yb + off);
      if (!LAST) pd[j] = ldg4(db + off);
    } else {
      const bool ok = s0 + row_of(j) < total;
      py[j] = ok ? ldg4(yb + off) : f32x4{0, 0, 0, 0};
      if (!LAST) pd[j] = ok ? ldg4(db + off) : f32x4{0, 0, 0, 0};
    }
  }
  __device__ __forceinline__ void load_groups(const BwdArgs &a, long long s0, long long total) {
    if (LAST) {
      const long long ra = s0 + row_of(0), rb = s0 + row_of(NPASS - 1);
      if (nA == -2) {  // first sub-tile of the workgroup
        nA = ra < total ? a.grow[ra] >> 6 : -1;
        nB = rb < total ? a.grow[rb] >> 6 : -1;
      }
      gA = __builtin_amdgcn_readfirstlane(nA);
      gB = __builtin_amdgcn_readfirstlane(nB < 0 ? nA : nB);
      nA = ra + kRows < total ? a.grow[ra + kRows] >> 6 : -1;
      nB = rb + kRows < total ? a.grow[rb + kRows] >> 6 : -1;
      // (d is scaled by a where it is used: the loads stay in flight under the MFMAs)
      if (gA >= 0) { dA = ldg4(a.d + static_cast<size_t>(gA) * COUT + 4 * cq); sA = ldg4i(a.sel + static_cast<size_t>(gA) * COUT + 4 * cq); }
      if (gB >= 0) { dB = ldg4(a.d + static_cast<size_t>(gB) * COUT + 4 * cq); sB = ldg4i(a.sel + static_cast<size_t>(gB) * COUT + 4 * cq); }
    }
  }
  template <bool FULL>
  __device__ __forceinline__ void prefetch(const BwdArgs &a, long long s0, long long total) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) load_row<FULL>(a, s0, total, j);
    load_groups(a, s0, total);
  }
  // the row words of this thread's rows, fetched together (one LDS round trip for the whole staging)
  struct RowWords {
    int packed[NPASS];
    float w[NPASS];
  };
  __device__ __forceinline__ RowWords row_words(const float *s_w, const int *s_grow) const {
    RowWords rw;
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      rw.packed[j] = s_grow[row_of(j)];
      rw.w[j] = s_w[row_of(j)];
    }
    return rw;
  }
  // dy of pass j (row row_of(j) of the sub-tile); zero for rows past the end (their group word is -1)
  __device__ __forceinline__ f32x4 value(const BwdArgs &a, int j, const RowWords &rw) {
    f32x4 dsel;
    if (LAST) {
      const int packed = __builtin_amdgcn_readfirstlane(rw.packed[j]);  // wave-uniform: the wave owns the row
      if (__builtin_expect(packed < 0, 0)) return f32x4{0, 0, 0, 0};
      const int g = packed >> 6, rin = packed & 63;
      f32x4 dg;
      i32x4 sg;
      if (__builtin_expect(g == gA, 1)) { dg = dA; sg = sA; }
      else if (__builtin_expect(g == gB, 1)) { dg = dB; sg = sB; }
      else { dg = ldg4(a.d + static_cast<size_t>(g) * COUT + 4 * cq); sg = ldg4i(a.sel + static_cast<size_t>(g) * COUT + 4 * cq); }
#pragma unroll
      for (int u = 0; u < 4; ++u) dsel[u] = rin == sg[u] ? ca[u] * dg[u] : 0.f;
    } else {
      if (rw.packed[j] < 0) return f32x4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) dsel[u] = ca[u] * pd[j][u];
    }
    return dy4(py[j], dsel, rw.w[j], cA, cB);
  }
};

template <int CIN, int COUT, bool LAST, bool FIRST>
__global__ __launch_bounds__(kT) void sa_bwd_dx_kernel(const BwdArgs a) {
  static_assert(!LAST || COUT == 256, "pooled layer: a wave stages whole rows");
  static_assert(CIN == 64 || CIN == 128, "wave tiling");
  constexpr int WN = CIN / 32;       // waves along the output columns (one 32-column block each)
  constexpr int WM = 4 / WN;         // waves along the rows
  constexpr int RBW = 2 / WM;        // 32-row blocks per wave
  constexpr int KK = COUT / 2;       // MFMA k-steps (contraction over the output channels of the layer)
  constexpr int SD = COUT + 4;
  // Wide layer (8 MFMAs per step): the next sub-tile's loads ride the matrix loop, one per step.  Narrow layer (4 per
  // step do not cover a load's issue): everything in front of the loop in the bounds-tested form -- whose exec-mask
  // branches happen to space the loads; the same loads back to back were slower (tools/sa_prof.py: 10.1 -> 12.8 k clocks).
  constexpr bool RIDE = RBW * 4 >= 8;
  extern __shared__ float lds[];
  float *s_dy = lds;                                 // [64][SD], channel c at (c & 1) * COUT/2 + (c >> 1)
  float *s_w = s_dy + kRows * SD;                    // [64]
  int *s_grow = reinterpret_cast<int *>(s_w + kRows);  // [64]
  float *s_x = reinterpret_cast<float *>(s_grow + kRows);  // FIRST: [64][4] grouped xyz of the rows
  constexpr int SI = CIN + 4;
  float *s_in = s_x;  // !FIRST: [64][SI] y_in of the sub-tile, overwritten in place by dmid_in (same LDS slot as s_x)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wn = wv / WM, wm = wv % WM;
  const Range rg = wg_range(a.goff, a.groups);
  const int kcol = 32 * wn + l31;  // this lane's input channel (column of dA)

  float acc_s = 0.f, acc_x = 0.f, acc_t0 = 0.f, acc_t1 = 0.f, acc_t2 = 0.f;
  if (rg.sub0 < rg.sub1) {
    // stationary operand: B[k = c = 2 cc + h][j = input channel] = w[c][kcol]
    float wreg[KK];
#pragma unroll
    for (int cc = 0; cc < KK; ++cc) wreg[cc] = a.w[static_cast<size_t>(2 * cc + h) * CIN + kcol];
    // the layer below, for the epilogue: scale, shift, mean, invstd of this lane's input channel
    const float e_sc = a.st_in[kcol], e_sh = a.st_in[CIN + kcol], e_mu = a.st_in[2 * CIN + kcol], e_is = a.st_in[3 * CIN + kcol];
    float e_w0 = 0.f, e_w1 = 0.f, e_w2 = 0.f;
    if (FIRST) { e_w0 = a.w1[3 * kcol]; e_w1 = a.w1[3 * kcol + 1]; e_w2 = a.w1[3 * kcol + 2]; }

    DyStage<COUT, LAST> st;
    st.init(a);
    float pw = 0.f, px[3] = {0.f, 0.f, 0.f};
    int pg = 0;
    constexpr int QI = CIN / 4, RI = kT / QI, NI = kRows / RI;  // staging of the y_in tile (16 bytes per lane)
    const int iq = tid % QI, ir0 = tid / QI;
    f32x4 pin[FIRST ? 1 : NI];
    auto prefetch = [&](long long sub) {
      const long long s0 = sub * kRows;
      if (RIDE && s0 + kRows <= rg.total) {  // a whole sub-tile (all but the last one): see DyStage::load_row
        st.template prefetch<true>(a, s0, rg.total);
        if (!FIRST) {
          const float *ib = a.src_in + s0 * CIN;
#pragma unroll
          for (int j = 0; j < NI; ++j) pin[j] = ldg4(ib + (ir0 + RI * j) * CIN + 4 * iq);
        }
        if (tid < kRows) {
          pw = (a.roww + s0)[tid];
          pg = (a.grow + s0)[tid];
          if (FIRST) {
            const float *xb = a.src_in + s0 * 3;
            px[0] = xb[tid * 3]; px[1] = xb[tid * 3 + 1]; px[2] = xb[tid * 3 + 2];
          }
        }
        return;
      }
      st.template prefetch<false>(a, s0, rg.total);
      if (!FIRST) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const long long r = s0 + ir0 + RI * j;
          pin[j] = r < rg.total ? ldg4(a.src_in + r * CIN + 4 * iq) : f32x4{0, 0, 0, 0};
        }
      }
      if (tid < kRows) {
        const long long r = s0 + tid;
        const bool ok = r < rg.total;
        pw = ok ? a.roww[r] : 0.f;
        pg = ok ? a.grow[r] : -1;
        if (FIRST) {
          px[0] = ok ? a.src_in[r * 3] : 0.f; px[1] = ok ? a.src_in[r * 3 + 1] : 0.f; px[2] = ok ? a.src_in[r * 3 + 2] : 0.f;
        }
      }
    };
    // the same loads for a WHOLE sub-tile, in pieces: piece i is issued from step i of the matrix loop (constant i)
    constexpr int NP = DyStage<COUT, LAST>::NPASS;
    constexpr int kParts = NP + (FIRST ? 0 : NI) + 2;
    static_assert(kParts <= KK / 4, "one piece per MFMA step");
    auto prefetch_part = [&](long long sub, int i) {
      const long long s0 = sub * kRows;
      if (i < NP) {
        st.template load_row<true>(a, s0, rg.total, i);
      } else if (!FIRST && i < NP + NI) {
        const float *ib = a.src_in + s0 * CIN;
        const int j = i - NP;
        pin[j] = ldg4(ib + (ir0 + RI * j) * CIN + 4 * iq);
      } else if (i == kParts - 2) {
        if (tid < kRows) {
          pw = (a.roww + s0)[tid];
          pg = (a.grow + s0)[tid];
          if (FIRST) {
            const float *xb = a.src_in + s0 * 3;
            px[0] = xb[tid * 3]; px[1] = xb[tid * 3 + 1]; px[2] = xb[tid * 3 + 2];
          }
        }
      } else if (i == kParts - 1) {
        st.load_groups(a, s0, rg.total);
      }
    };
    prefetch(rg.sub0);
    SA_PROF_DECL;
    for (long long sub = rg.sub0; sub < rg.sub1; ++sub) {
      const long long s0 = sub * kRows;
      lds_barrier();
      SA_PROF_MARK(0);
      SA_PROF_COUNT(9, 1);
      if (tid < kRows) {
        s_w[tid] = pw; s_grow[tid] = pg;
        if (FIRST) { s_x[4 * tid] = px[0]; s_x[4 * tid + 1] = px[1]; s_x[4 * tid + 2] = px[2]; s_x[4 * tid + 3] = 0.f; }
      }
      if (!FIRST) {
#pragma unroll
        for (int j = 0; j < NI; ++j) *reinterpret_cast<f32x4 *>(s_in + (ir0 + RI * j) * SI + 4 * iq) = pin[j];
      }
      lds_barrier();
      SA_PROF_MARK(1);
      {
        const auto rw = st.row_words(s_w, s_grow);
#pragma unroll
        for (int j = 0; j < DyStage<COUT, LAST>::NPASS; ++j) {
          const f32x4 v = st.value(a, j, rw);
          const int row = st.row_of(j);
          *reinterpret_cast<f32x2 *>(s_dy + row * SD + 2 * st.cq) = f32x2{v[0], v[2]};
          *reinterpret_cast<f32x2 *>(s_dy + row * SD + COUT / 2 + 2 * st.cq) = f32x2{v[1], v[3]};
        }
      }
      SA_PROF_MARK(2);
      lds_barrier();
      SA_PROF_MARK(3);
      const bool more = sub + 1 < rg.sub1;
      const bool whole = RIDE && more && (sub + 2) * kRows <= rg.total;  // the next sub-tile is a whole one: loads ride the loop
      if (__builtin_expect(more && !whole, 0)) prefetch(sub + 1);

      f32x16 acc[RBW];
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
      // fragments of step kq + 1 are fetched before the MFMAs of step kq are issued (two register sets): left to itself
      // the compiler reuses one set and every 8 RBW MFMAs wait a full LDS round trip (measured: 88 instead of 64 clocks
      // per MFMA, tools/sa_prof.py)
      f32x4 av[2][RBW];
      auto frag = [&](int kq, int set) {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
          av[set][rb] = *reinterpret_cast<const f32x4 *>(s_dy + (32 * (wm * RBW + rb) + l31) * SD + h * (COUT / 2) + 4 * kq);
      };
      frag(0, 0);
#pragma unroll
      for (int kq = 0; kq < KK / 4; ++kq) {
        if (kq + 1 < KK / 4) frag(kq + 1, (kq + 1) & 1);
        if (kq < kParts && __builtin_expect(whole, 1)) prefetch_part(sub + 1, kq);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int rb = 0; rb < RBW; ++rb) acc[rb] = mfma(av[kq & 1][rb][i], wreg[4 * kq + i], acc[rb]);
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- epilogue: ReLU mask of the layer below, its BN-backward sums, store
      SA_PROF_MARK(4);
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * (wm * RBW + rb) + (e & 3) + 8 * (e >> 2) + 4 * h;
          const long long r = s0 + row;
          float y;  // y_in of this accumulator element
          if (FIRST) {
            const f32x4 xr = *reinterpret_cast<const f32x4 *>(s_x + 4 * row);
            y = dot3w(xr[0], xr[1], xr[2], e_w0, e_w1, e_w2);
          } else {
            y = s_in[row * SI + kcol];
          }
          const bool on = __fadd_rn(__fmul_rn(y, e_sc), e_sh) > 0.0f && r < rg.total;
          const float dm = on ? acc[rb][e] : 0.f;
          acc_s += dm;
          acc_x = fmaf(dm, (y - e_mu) * e_is, acc_x);
          if (FIRST) {
            const f32x4 xr = *reinterpret_cast<const f32x4 *>(s_x + 4 * row);
            acc_t0 = fmaf(dm, xr[0], acc_t0); acc_t1 = fmaf(dm, xr[1], acc_t1); acc_t2 = fmaf(dm, xr[2], acc_t2);
          } else {
            s_in[row * SI + kcol] = dm;  // in place: this lane is the only reader / writer of the element
          }
        }
      SA_PROF_MARK(5);
      if (!FIRST) {  // dmid_in leaves as whole rows, 16 bytes per lane
        lds_barrier();
        SA_PROF_MARK(6);
        if (__builtin_expect(s0 + kRows <= rg.total, 1)) {  // (uniform) a whole sub-tile: read from LDS in one go, then stored
          float *ob = a.dmid_in + s0 * CIN;
          f32x4 t[NI];
#pragma unroll
          for (int j = 0; j < NI; ++j) t[j] = *reinterpret_cast<const f32x4 *>(s_in + (ir0 + RI * j) * SI + 4 * iq);
#pragma unroll
          for (int j = 0; j < NI; ++j) *reinterpret_cast<f32x4 *>(ob + (ir0 + RI * j) * CIN + 4 * iq) = t[j];
        } else {
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int row = ir0 + RI * j;
            if (s0 + row < rg.total)
              *reinterpret_cast<f32x4 *>(a.dmid_in + (s0 + row) * CIN + 4 * iq) = *reinterpret_cast<const f32x4 *>(s_in + row * SI + 4 * iq);
          }
        }
      }
      SA_PROF_MARK(7);
    }
    SA_PROF_STORE(LAST ? 2 : 3);
  }
  // sums of this lane's input channel: the two half-waves hold different rows
  acc_s += __shfl_xor(acc_s, 32);
  acc_x += __shfl_xor(acc_x, 32);
  if (FIRST) { acc_t0 += __shfl_xor(acc_t0, 32); acc_t1 += __shfl_xor(acc_t1, 32); acc_t2 += __shfl_xor(acc_t2, 32); }
  if (h == 0 && rg.sub0 < rg.sub1) {
    atomicAdd(a.sums_in + kcol, static_cast<double>(acc_s));
    atomicAdd(a.sums_in + CIN + kcol, static_cast<double>(acc_x));
    if (FIRST) {
      atomicAdd(a.sums_in + 2 * CIN + kcol, static_cast<double>(acc_t0));
      atomicAdd(a.sums_in + 3 * CIN + kcol, static_cast<double>(acc_t1));
      atomicAdd(a.sums_in + 4 * CIN + kcol, static_cast<double>(acc_t2));
    }
  }
}

template <int CIN, int COUT, bool LAST, bool FIRST>
__global__ __launch_bounds__(kT) void sa_bwd_dw_kernel(const BwdArgs a) {
  static_assert(!LAST || COUT == 256, "pooled layer: a wave stages whole rows");
  constexpr int IB = COUT / 32, JB = CIN / 32;
  static_assert(IB % 4 == 0, "wave tiling");
  constexpr int IBW = IB / 4;        // 32-row blocks of dW per wave (all JB column blocks)
  constexpr int SD = COUT;           // dy tile [64][COUT], natural layout
  constexpr int SA = FIRST ? CIN + 16 : CIN;  // act tile [64][SA]: FIRST's writers hold a quarter row each (4 rows per
                                              // 16-lane group): 16 floats of padding spread the rows over the banks
  constexpr int QPA = CIN / 4, RPA = kT / QPA, NPA = kRows / RPA;
  constexpr bool RIDE = IBW * JB >= 8;  // as in the dx kernel (narrow layer: 2 MFMAs per step)
  extern __shared__ float lds[];
  float *s_dy = lds;
  float *s_a = s_dy + kRows * SD;
  float *s_w = s_a + kRows * SA;
  int *s_grow = reinterpret_cast<int *>(s_w + kRows);
  float *s_w1 = reinterpret_cast<float *>(s_grow + kRows);  // FIRST: [CIN][4] w1, [CIN][2] scale / shift

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const Range rg = wg_range(a.goff, a.groups);

  f32x16 acc[IBW][JB];
#pragma unroll
  for (int ib = 0; ib < IBW; ++ib)
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ib][jb][e] = 0.f;

  if (rg.sub0 < rg.sub1) {
    if (FIRST) {
      for (int k = tid; k < CIN; k += kT) {
        s_w1[4 * k] = a.w1[3 * k]; s_w1[4 * k + 1] = a.w1[3 * k + 1]; s_w1[4 * k + 2] = a.w1[3 * k + 2]; s_w1[4 * k + 3] = 0.f;
        s_w1[4 * CIN + 2 * k] = a.st_in[k]; s_w1[4 * CIN + 2 * k + 1] = a.st_in[CIN + k];
      }
    }
    DyStage<COUT, LAST> st;
    st.init(a);
    const int aq = tid % QPA, ar0 = tid / QPA;
    f32x4 sc4 = {0, 0, 0, 0}, sh4 = {0, 0, 0, 0};
    if (!FIRST) { sc4 = ldg4(a.st_in + 4 * aq); sh4 = ldg4(a.st_in + CIN + 4 * aq); }
    const int frow = tid >> 2, fpart = tid & 3;
    f32x4 pa[FIRST ? 1 : NPA];
    float px0 = 0.f, px1 = 0.f, px2 = 0.f, pw = 0.f;
    int pg = 0;
    auto prefetch = [&](long long sub) {
      const long long s0 = sub * kRows;
      if (RIDE && s0 + kRows <= rg.total) {  // a whole sub-tile (all but the last one): see DyStage::load_row
        st.template prefetch<true>(a, s0, rg.total);
        if (FIRST) {
          const float *xb = a.src_in + s0 * 3;
          px0 = xb[frow * 3]; px1 = xb[frow * 3 + 1]; px2 = xb[frow * 3 + 2];
        } else {
          const float *ab = a.src_in + s0 * CIN;
#pragma unroll
          for (int j = 0; j < NPA; ++j) pa[j] = ldg4(ab + (ar0 + RPA * j) * CIN + 4 * aq);
        }
        if (tid < kRows) {
          pw = (a.roww + s0)[tid];
          pg = (a.grow + s0)[tid];
        }
        return;
      }
      st.template prefetch<false>(a, s0, rg.total);
      if (FIRST) {
        const long long r = s0 + frow;
        const bool ok = r < rg.total;
        px0 = ok ? a.src_in[r * 3] : 0.f; px1 = ok ? a.src_in[r * 3 + 1] : 0.f; px2 = ok ? a.src_in[r * 3 + 2] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
          const long long r = s0 + ar0 + RPA * j;
          pa[j] = r < rg.total ? ldg4(a.src_in + r * CIN + 4 * aq) : f32x4{0, 0, 0, 0};
        }
      }
      if (tid < kRows) {
        const long long r = s0 + tid;
        pw = r < rg.total ? a.roww[r] : 0.f;
        pg = r < rg.total ? a.grow[r] : -1;
      }
    };
    // the same loads for a WHOLE sub-tile, in pieces: piece i is issued from step i of the matrix loop (constant i)
    constexpr int NP = DyStage<COUT, LAST>::NPASS;
    constexpr int kParts = NP + (FIRST ? 1 : NPA) + 2;
    static_assert(kParts <= kRows / 2, "one piece per MFMA step");
    auto prefetch_part = [&](long long sub, int i) {
      const long long s0 = sub * kRows;
      if (i < NP) {
        st.template load_row<true>(a, s0, rg.total, i);
      } else if (i < kParts - 2) {
        if (FIRST) {
          const float *xb = a.src_in + s0 * 3;
          px0 = xb[frow * 3]; px1 = xb[frow * 3 + 1]; px2 = xb[frow * 3 + 2];
        } else {
          const float *ab = a.src_in + s0 * CIN;
          const int j = i - NP;
          pa[j] = ldg4(ab + (ar0 + RPA * j) * CIN + 4 * aq);
        }
      } else if (i == kParts - 2) {
        if (tid < kRows) {
          pw = (a.roww + s0)[tid];
          pg = (a.grow + s0)[tid];
        }
      } else if (i == kParts - 1) {
        st.load_groups(a, s0, rg.total);
      }
    };
    prefetch(rg.sub0);
    SA_PROF_DECL;
    for (long long sub = rg.sub0; sub < rg.sub1; ++sub) {
      const long long s0 = sub * kRows;
      lds_barrier();
      SA_PROF_MARK(0);
      SA_PROF_COUNT(9, 1);
      if (tid < kRows) { s_w[tid] = pw; s_grow[tid] = pg; }
      // activations of the layer below
      if (FIRST) {
        const float okf = s0 + frow < rg.total ? 1.0f : 0.0f;  // (a multiplication, not a branch: see the forward kernel)
#pragma unroll
        for (int i = 0; i < CIN / 4; i += 4) {
          const int k0 = 4 * i + 4 * fpart;  // (see the forward kernel; with SA = CIN + 16 the b128 writes are conflict-free)
          f32x4 v;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 wk = *reinterpret_cast<const f32x4 *>(s_w1 + 4 * (k0 + u));
            const f32x2 ss = *reinterpret_cast<const f32x2 *>(s_w1 + 4 * CIN + 2 * (k0 + u));
            v[u] = okf * bn_act(dot3w(px0, px1, px2, wk[0], wk[1], wk[2]), ss[0], ss[1]);
          }
          *reinterpret_cast<f32x4 *>(s_a + frow * SA + k0) = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
          const int row = ar0 + RPA * j;
          const bool ok = s0 + row < rg.total;
          f32x4 v;
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = ok ? bn_act(pa[j][u], sc4[u], sh4[u]) : 0.f;
          *reinterpret_cast<f32x4 *>(s_a + row * SA + 4 * aq) = v;
        }
      }
      SA_PROF_MARK(1);
      lds_barrier();
      SA_PROF_MARK(2);
      {
        const auto rw = st.row_words(s_w, s_grow);
#pragma unroll
        for (int j = 0; j < DyStage<COUT, LAST>::NPASS; ++j) {
          const f32x4 v = st.value(a, j, rw);
          *reinterpret_cast<f32x4 *>(s_dy + st.row_of(j) * SD + 4 * st.cq) = v;
        }
      }
      SA_PROF_MARK(3);
      lds_barrier();
      SA_PROF_MARK(4);
      const bool more = sub + 1 < rg.sub1;
      const bool whole = RIDE && more && (sub + 2) * kRows <= rg.total;  // the next sub-tile is a whole one: loads ride the loop
      if (__builtin_expect(more && !whole, 0)) prefetch(sub + 1);

      // dW[c][k] += sum_rows dy[row][c] act[row][k]: A[i = c][k = row], B[k = row][j = k-channel]
      // two fragment register sets, as in the dx kernel: step kk + 1 is fetched before step kk's MFMAs are issued
      float af[2][IBW], bf[2][JB];
      auto frag = [&](int kk, int set) {
#pragma unroll
        for (int ib = 0; ib < IBW; ++ib) af[set][ib] = s_dy[(2 * kk + h) * SD + 32 * (IBW * wv + ib) + l31];
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) bf[set][jb] = s_a[(2 * kk + h) * SA + 32 * jb + l31];
      };
      frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < kRows / 2; ++kk) {
        if (kk + 1 < kRows / 2) frag(kk + 1, (kk + 1) & 1);
        if (kk < kParts && __builtin_expect(whole, 1)) prefetch_part(sub + 1, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ib = 0; ib < IBW; ++ib)
#pragma unroll
          for (int jb = 0; jb < JB; ++jb) acc[ib][jb] = mfma(af[kk & 1][ib], bf[kk & 1][jb], acc[ib][jb]);
        __builtin_amdgcn_sched_barrier(0);
      }
      SA_PROF_MARK(5);
    }
    SA_PROF_STORE(LAST ? 4 : 5);
  }
  // this workgroup's partial tile (zeros when it had no rows)
  float *out = a.partials + static_cast<size_t>(blockIdx.x) * COUT * CIN;
#pragma unroll
  for (int ib = 0; ib < IBW; ++ib)
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = 32 * (IBW * wv + ib) + (e & 3) + 8 * (e >> 2) + 4 * h;
        out[static_cast<size_t>(c) * CIN + 32 * jb + l31] = acc[ib][jb][e];
      }
}

// dw[i] = sum over the workgroups' partial tiles in a fixed order (deterministic): a workgroup owns 64 outputs, its 4
// waves take a quarter of the tiles each (8 loads in flight per lane), the quarters are added in order
__global__ __launch_bounds__(kT) void dw_reduce_kernel(const float *__restrict__ partials, float *__restrict__ dw, int n, int nblk) {
  __shared__ float s_p[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int per = (nblk + 3) / 4, b0 = q * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  float s = 0.f;
  if (i < n) {
    const float *p = partials + i;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[static_cast<size_t>(b + u) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += p[static_cast<size_t>(b) * n];
  }
  s_p[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && i < n) dw[i] = ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x];
}

int device_cus() {
  static std::mutex mu;
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
  std::lock_guard<std::mutex> lock(mu);
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cached[dev] = n;
  }
  return cached[dev];
}

template <int CIN, int COUT, bool FIRST, bool POOL>
int launch_fwd(const FwdArgs &a, int nblk, hipStream_t s) {
  size_t lds = sizeof(float) * ((fwd_alias(CIN, COUT) ? 0 : kRows * (CIN + 4)) + kRows * (COUT + 4) + 2 * kRows);
  if (FIRST) lds += sizeof(float) * 6 * CIN;
  auto kern = sa_fwd_kernel<CIN, COUT, FIRST, POOL>;
  int st = raise_dynamic_lds(kern, lds);
  if (st != CODA_OK) return st;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(kT), lds, s, a);
  return launch_status();
}
template <int CIN, int COUT, bool LAST, bool FIRST>
int launch_dx(const BwdArgs &a, int nblk, hipStream_t s) {
  size_t lds = sizeof(float) * (kRows * (COUT + 4) + 2 * kRows + (FIRST ? 4 * kRows : kRows * (CIN + 4)));
  auto kern = sa_bwd_dx_kernel<CIN, COUT, LAST, FIRST>;
  int st = raise_dynamic_lds(kern, lds);
  if (st != CODA_OK) return st;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(kT), lds, s, a);
  return launch_status();
}
template <int CIN, int COUT, bool LAST, bool FIRST>
int launch_dw(const BwdArgs &a, int nblk, hipStream_t s) {
  size_t lds = sizeof(float) * (kRows * COUT + kRows * (FIRST ? CIN + 16 : CIN) + 2 * kRows + (FIRST ? 6 * CIN : 0));
  auto kern = sa_bwd_dw_kernel<CIN, COUT, LAST, FIRST>;
  int st = raise_dynamic_lds(kern, lds);
  if (st != CODA_OK) return st;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(kT), lds, s, a);
  return launch_status();
}

bool fill_coef(BwdArgs &a, const float *coef, int layout, int cout) {
  if (!coef) return false;
  if (layout == 1) {  // [a, m1, m2, mean, invstd]
    a.ca = coef; a.cm1 = coef + cout; a.cm2 = coef + 2 * cout; a.cmean = coef + 3 * cout; a.cinv = coef + 4 * cout;
  } else if (layout == 0) {  // [scale, shift, mean, invstd, a, m1, m2]
    a.cmean = coef + 2 * cout; a.cinv = coef + 3 * cout; a.ca = coef + 4 * cout; a.cm1 = coef + 5 * cout; a.cm2 = coef + 6 * cout;
  } else {
    return false;
  }
  return true;
}

}  // namespace
}  // namespace coda

using namespace coda;

// kind 0: forward / dx kernels -- 2 row ranges per CU, dealt to the CUs as they become free (a range costs its
// workgroup only the weight fragments, ~2 % of its time; with exactly one workgroup per CU a concurrent kernel that
// holds a few CUs -- the sampling of the next batch on its side stream -- doubled the kernel's duration: 8 of the 256
// workgroups had to wait for a whole pass of the others).  kind 1: dw kernels -- one per CU (a 128 KB partial tile each).
#ifdef CODA_SA_PROF
CODA_API int coda_sa_prof_read(unsigned long long *host) {
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(coda::g_sa_prof), sizeof(coda::g_sa_prof)));
}
#endif

CODA_API int coda_sa_mfma_blocks(int kind) { return kind == 0 ? 2 * device_cus() : device_cus(); }

CODA_API int coda_sa_mfma_supported(int c1, int c2, int c3, int s_len) {
  return c1 == 64 && c2 == 128 && c3 == 256 && s_len >= 1 && s_len <= kRows ? 1 : 0;
}

CODA_API int coda_sa_pack_groups_f32(const float *grouped, const int32_t *idx, int dedup, float *x, float *row_weight,
                                     int32_t *group_offsets, int32_t *row_group, double *moments, int32_t *counts,
                                     double *zero, int nzero, long long groups, int s_len, void *stream) {
  // row_group packs (group << 6) | row-in-group into an int32: at most 2^25 groups of at most 64 rows
  if (groups < 0 || groups >= (1LL << 25) || s_len <= 0 || s_len > kRows || nzero < 0 || groups * s_len > 0x7fffffffLL)
    return CODA_EINVAL;
  if (!group_offsets || !moments || (nzero > 0 && !zero)) return CODA_EINVAL;
  if (groups > 0 && (!grouped || !idx || !x || !row_weight || !row_group || !counts)) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  if (groups > 0)
    hipLaunchKernelGGL(pack_count_kernel, dim3(static_cast<unsigned>((groups + 3) / 4)), dim3(kT), 0, s, idx, counts, groups,
                       s_len, dedup);
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, s, counts, group_offsets, groups, moments, zero, nzero);
  if (groups > 0)
  {
    long long blocks = (groups * s_len + kT - 1) / kT;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(pack_scatter_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kT), 0, s, grouped, counts,
                       group_offsets, x, row_weight, row_group, moments, groups, s_len);
  }
  return launch_status();
}

CODA_API int coda_sa_l1_sums_f32(const double *moments, const float *w1, double *sums, int c, void *stream) {
  if (c <= 0 || !moments || !w1 || !sums) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(l1_sums_kernel, dim3((c + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), moments, w1, sums, c);
  return launch_status();
}

CODA_API int coda_sa_mfma_fwd_f32(const float *src, const float *w1, const float *stats_in, const float *w,
                                  const float *row_weight, const int32_t *group_offsets, const int32_t *row_group,
                                  long long groups, int s_len, int cin, int cout, float *y_out, double *sums,
                                  const float *gamma, float *ysel, int32_t *sel, float *part_y, int32_t *part_sel,
                                  int32_t *part_gid, int nblocks, void *stream) {
  if (groups < 0 || s_len <= 0 || s_len > kRows || nblocks <= 0) return CODA_EINVAL;
  if (groups == 0) return CODA_OK;
  if (!src || !stats_in || !w || !row_weight || !group_offsets || !row_group || !sums) return CODA_EINVAL;
  const bool pool = ysel != nullptr;
  if (pool && (!gamma || !sel || !part_y || !part_sel || !part_gid)) return CODA_EINVAL;
  FwdArgs a{src, w1, stats_in, w, row_weight, group_offsets, row_group, groups, y_out, sums, gamma, ysel, sel, part_y, part_sel, part_gid};
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_sticky_error();
  if (w1 && !pool && cin == 64 && cout == 128) return launch_fwd<64, 128, true, false>(a, nblocks, s);
  if (!w1 && pool && cin == 128 && cout == 256) return launch_fwd<128, 256, false, true>(a, nblocks, s);
  return CODA_EINVAL;
}

CODA_API int coda_sa_pool_finish_f32(float *ysel, int32_t *sel, const float *part_y, const int32_t *part_sel,
                                     const int32_t *part_gid, const int32_t *group_offsets, const float *gamma,
                                     const float *stats, float *out, long long groups, int c, int nblocks, void *stream) {
  if (groups < 0 || c <= 0 || nblocks <= 0) return CODA_EINVAL;
  if (groups == 0) return CODA_OK;
  if (!ysel || !sel || !part_y || !part_sel || !part_gid || !group_offsets || !gamma || !stats || !out) return CODA_EINVAL;
  long long blocks = (groups * c + kT * 4 - 1) / (kT * 4);
  blocks = blocks > 8192 ? 8192 : blocks;
  clear_sticky_error();
  hipLaunchKernelGGL(pool_finish_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kT), 0, static_cast<hipStream_t>(stream),
                     ysel, sel, part_y, part_sel, part_gid, group_offsets, gamma, stats, out, groups, c, nblocks);
  return launch_status();
}

namespace {
int bwd_args(BwdArgs &a, const float *y_out, const float *dmid, const float *d, const int32_t *sel, const float *coef,
             int layout, const float *src_in, const float *w1, const float *stats_in, const float *row_weight,
             const int32_t *group_offsets, const int32_t *row_group, long long groups, int s_len, int cout, int nblocks) {
  if (groups < 0 || s_len <= 0 || s_len > kRows || nblocks <= 0) return CODA_EINVAL;
  if (!y_out || !src_in || !stats_in || !row_weight || !group_offsets || !row_group) return CODA_EINVAL;
  const bool last = d != nullptr;
  if (last ? (!sel || dmid || layout != 1) : (!dmid || layout != 0)) return CODA_EINVAL;
  a = BwdArgs{};
  a.y_out = y_out; a.dmid = dmid; a.d = d; a.sel = sel;
  if (!fill_coef(a, coef, layout, cout)) return CODA_EINVAL;
  a.src_in = src_in; a.w1 = w1; a.st_in = stats_in; a.roww = row_weight; a.goff = group_offsets; a.grow = row_group;
  a.groups = groups;
  return CODA_OK;
}
}  // namespace

CODA_API int coda_sa_mfma_bwd_dx_f32(const float *y_out, const float *dmid, const float *d, const int32_t *sel,
                                     const float *coef, int layout, const float *w, const float *src_in, const float *w1,
                                     const float *stats_in, const float *row_weight, const int32_t *group_offsets,
                                     const int32_t *row_group, long long groups, int s_len, int cin, int cout,
                                     float *dmid_in, double *sums_in, int nblocks, void *stream) {
  if (!sums_in || cin <= 0) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums_in, 0, sizeof(double) * (w1 ? 5 : 2) * cin, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (groups == 0) return CODA_OK;
  BwdArgs a;
  int st = bwd_args(a, y_out, dmid, d, sel, coef, layout, src_in, w1, stats_in, row_weight, group_offsets, row_group,
                    groups, s_len, cout, nblocks);
  if (st != CODA_OK) return st;
  if (!w || (!w1 && !dmid_in)) return CODA_EINVAL;
  a.w = w; a.dmid_in = dmid_in; a.sums_in = sums_in;
  clear_sticky_error();
  if (d && !w1 && cin == 128 && cout == 256) return launch_dx<128, 256, true, false>(a, nblocks, s);
  if (!d && w1 && cin == 64 && cout == 128) return launch_dx<64, 128, false, true>(a, nblocks, s);
  return CODA_EINVAL;
}

CODA_API int coda_sa_mfma_bwd_dw_f32(const float *y_out, const float *dmid, const float *d, const int32_t *sel,
                                     const float *coef, int layout, const float *src_in, const float *w1,
                                     const float *stats_in, const float *row_weight, const int32_t *group_offsets,
                                     const int32_t *row_group, long long groups, int s_len, int cin, int cout,
                                     float *partials, float *dw, int nblocks, void *stream) {
  if (!dw || !partials || cin <= 0 || cout <= 0) return CODA_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (groups == 0) return static_cast<int>(hipMemsetAsync(dw, 0, sizeof(float) * cin * cout, s));
  BwdArgs a;
  int st = bwd_args(a, y_out, dmid, d, sel, coef, layout, src_in, w1, stats_in, row_weight, group_offsets, row_group,
                    groups, s_len, cout, nblocks);
  if (st != CODA_OK) return st;
  a.partials = partials;
  clear_sticky_error();
  if (d && !w1 && cin == 128 && cout == 256) st = launch_dw<128, 256, true, false>(a, nblocks, s);
  else if (!d && w1 && cin == 64 && cout == 128) st = launch_dw<64, 128, false, true>(a, nblocks, s);
  else return CODA_EINVAL;
  if (st != CODA_OK) return st;
  const int n = cin * cout;
  hipLaunchKernelGGL(dw_reduce_kernel, dim3((n + 63) / 64), dim3(kT), 0, s, partials, dw, n, nblocks);
  return launch_status();
}

CODA_API int coda_sa_l1_bwd_f32(const double *sums5, const double *sums_bn, double n, const float *gamma, const float *stats,
                                const double *moments, const float *w1, float *dw1, float *dbeta, float *dgamma, int c,
                                void *stream) {
  if (c <= 0 || !sums5) return CODA_EINVAL;
  if (dw1 && (!sums_bn || !gamma || !stats || !moments || !w1)) return CODA_EINVAL;
  clear_sticky_error();
  hipLaunchKernelGGL(l1_bwd_kernel, dim3((c + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sums5, sums_bn, n,
                     gamma, stats, moments, w1, dw1, dbeta, dgamma, c);
  return launch_status();
}
