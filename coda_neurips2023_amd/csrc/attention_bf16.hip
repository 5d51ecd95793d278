// attention_bf16.hip -- the fused attention core with bf16 MFMA operands (v_mfma_f32_32x32x16_bf16).
//
// BASELINE.json configs[4] (ScanNet-shaped: 40k points, 512 queries, "bf16 MFMA attention"): the same
// three kernels as attention.hip -- forward, dK/dV, dQ -- for the same fp32 tensors, with Q, K, V, dO and
// the probabilities rounded to bf16 on their way into the matrix cores; accumulation, the online
// softmax, lse, delta and every output stay fp32.  16x the MFMA rate of the fp32 path, so these kernels
// are bound by the softmax VALU work and the LDS traffic instead.
//
// Fragment plan (32x32x16: A[i = lane&31][8 k-slots of half = lane>>5], B[8 k-slots][j = lane&31],
// C/D as the fp32 32x32 forms: col = lane&31, row = crow(reg, half)).  The mapping of the 16 k-slots to
// contraction indices is free as long as A and B agree, which is what makes the second GEMM of each
// kernel cheap: the 16 accumulator registers of the first GEMM (rows crow(r, half)) are converted in
// place, registers 8jj..8jj+7 becoming the 8 slots of k-step jj, and the other operand is read from a
// TRANSPOSED bf16 tile in LDS at exactly those row numbers: rows 16jj + 4*half + {0..3} and
// 16jj + 8 + 4*half + {0..3}, i.e. two ds_read_b64 per fragment.  Tiles are therefore staged twice where
// a kernel needs both roles: row-major [32][D] (row stride 2D + 16 B: conflict-free ds_read_b128) and
// transposed [D][32] (row stride 72 B: conflict-free ds_read_b64).  Both images come from the same
// global loads: a thread fetches a 4 x 4 block (four float4 of four consecutive rows) and writes it
// once by rows and once by columns, 8 B per store.
//
// Staging: one LDS stage plus a register prefetch -- the global loads of stage i+1 are issued before the
// MFMAs of stage i and converted / stored after a barrier; the small LDS footprint (37-74 KB) leaves
// room for two workgroups per CU, which is what overlaps one workgroup's barrier with the other's math.
#include "attention_common.hip.h"

#include <algorithm>

namespace coda {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;  // 4 waves

template <int D>
struct Lay {
  static constexpr int RS = 2 * D + 16;  // bytes per row, row-major tile [32][D]
  static constexpr int TS = 72;          // bytes per row, transposed tile [D][32]
  static constexpr int ROWB = kTile * RS;
  static constexpr int TRB = D * TS;
};

__device__ __forceinline__ bf16x4 cvt4(float a, float b, float c, float d) {
  const f32x4 v = {a, b, c, d};
  return __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ bf16x8 cvt8(float a, float b, float c, float d, float e, float f, float g, float h) {
  const f32x8 t = {a, b, c, d, e, f, g, h};
  return __builtin_convertvector(t, bf16x8);
}
// registers 8jj .. 8jj+7 of a 16-element accumulator image (by value: the arrays must stay in VGPRs)
#define CODA_CVT8(arr, jj)                                                                              \
  cvt8(arr[8 * (jj)], arr[8 * (jj) + 1], arr[8 * (jj) + 2], arr[8 * (jj) + 3], arr[8 * (jj) + 4],       \
       arr[8 * (jj) + 5], arr[8 * (jj) + 6], arr[8 * (jj) + 7])
__device__ __forceinline__ bf16x8 join(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// NS = 1: operands rounded to bf16 (configs[4]).  NS = 3: every fp32 operand is carried as THREE bf16 pieces
// x = hi + mid + lo (8 + 8 + 8 significand bits: the fp32 value exactly, each residual is computed without
// rounding error) and a product a.b is evaluated as the six piece products of order <= 2 -- hi.hi, hi.mid,
// mid.hi, mid.mid, hi.lo, lo.hi -- into the fp32 accumulator; what is dropped (mid.lo, lo.mid, lo.lo) is
// <= 2^-24 relative, the size of fp32's own product rounding.  Six bf16 MFMAs cost 6/16 of one fp32 MFMA, and,
// unlike the fp32 MFMA, they run on the matrix cores next to the soft-max VALU work.
template <int NS>
struct Frag {
  bf16x8 v[NS];
};
template <int NS>
struct Frag4 {
  bf16x4 v[NS];
};
template <int NS>
__device__ __forceinline__ Frag<NS> split8(f32x8 x) {
  Frag<NS> f;
  f.v[0] = __builtin_convertvector(x, bf16x8);
  if constexpr (NS == 3) {
    x = x - __builtin_convertvector(f.v[0], f32x8);
    f.v[1] = __builtin_convertvector(x, bf16x8);
    x = x - __builtin_convertvector(f.v[1], f32x8);
    f.v[2] = __builtin_convertvector(x, bf16x8);
  }
  return f;
}
template <int NS>
__device__ __forceinline__ Frag4<NS> split4(f32x4 x) {
  Frag4<NS> f;
  f.v[0] = __builtin_convertvector(x, bf16x4);
  if constexpr (NS == 3) {
    x = x - __builtin_convertvector(f.v[0], f32x4);
    f.v[1] = __builtin_convertvector(x, bf16x4);
    x = x - __builtin_convertvector(f.v[1], f32x4);
    f.v[2] = __builtin_convertvector(x, bf16x4);
  }
  return f;
}
// registers 8jj .. 8jj+7 of a 16-element accumulator image as an operand
#define CODA_SPLIT8(NS, arr, jj)                                                                          \
  split8<NS>(f32x8{arr[8 * (jj)], arr[8 * (jj) + 1], arr[8 * (jj) + 2], arr[8 * (jj) + 3], arr[8 * (jj) + 4], \
                   arr[8 * (jj) + 5], arr[8 * (jj) + 6], arr[8 * (jj) + 7]})

// Fragment of a transposed tile for k-step jj: row `row` (a head-dim or feature index), the 8 slots of
// this lane's half (see the header comment).
// (`img`: bytes between the hi / mid / lo images of the tile set)
template <int D, int NS>
__device__ __forceinline__ Frag<NS> read_tr(const unsigned char *tile, int img, int row, int jj, int half) {
  const unsigned char *p = tile + row * Lay<D>::TS + (16 * jj + 4 * half) * 2;
  Frag<NS> f;
#pragma unroll
  for (int i = 0; i < NS; ++i)
    f.v[i] = join(*reinterpret_cast<const bf16x4 *>(p + i * img), *reinterpret_cast<const bf16x4 *>(p + i * img + 16));
  return f;
}
// Fragment of a row-major tile for k-step c: row `row`, head-dim components 16c + 8*half .. +7.
template <int D, int NS>
__device__ __forceinline__ Frag<NS> read_rm(const unsigned char *tile, int img, int row, int c, int half) {
  Frag<NS> f;
#pragma unroll
  for (int i = 0; i < NS; ++i)
    f.v[i] = *reinterpret_cast<const bf16x8 *>(tile + i * img + row * Lay<D>::RS + (16 * c + 8 * half) * 2);
  return f;
}
// The same fragment straight from an fp32 row in global memory (operands that stay in registers).
template <int NS>
__device__ __forceinline__ Frag<NS> load_frag(const float *row, bool valid, float scale) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (valid) {
    a = *reinterpret_cast<const float4 *>(row);
    b = *reinterpret_cast<const float4 *>(row + 4);
  }
  return split8<NS>(f32x8{a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale,
                          b.w * scale});
}

// NTILES tiles of 32 rows x D fp32, fetched by the 256 threads as 4 x 4 blocks (see the header).
template <int D, int NTILES, int THREADS = kThreads>
struct Fetch {
  static constexpr int CB = D / 4, BLK = NTILES * 8 * CB, PER = (BLK + THREADS - 1) / THREADS;
  static_assert(BLK % THREADS == 0 || BLK < THREADS, "tile set must split evenly over the workgroup");
  static constexpr bool kPartial = BLK < THREADS;  // one tile, 256 threads: the upper half of the workgroup idles
  float4 v[PER][4];

  __device__ __forceinline__ void load(const float *g, size_t gstride, int row0, int nrows, int tid) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int blk = tid + u * THREADS;
      if (kPartial && blk >= BLK) continue;
      const int tile = blk / (8 * CB), rb = (blk / CB) % 8, cb = blk % CB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + tile * kTile + 4 * rb + i;
        v[u][i] = row < nrows ? *reinterpret_cast<const float4 *>(g + static_cast<size_t>(row) * gstride + 4 * cb)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  template <int NS>
  __device__ __forceinline__ void store_rm(unsigned char *lds, int img, int tid) const {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int blk = tid + u * THREADS;
      if (kPartial && blk >= BLK) continue;
      const int tile = blk / (8 * CB), rb = (blk / CB) % 8, cb = blk % CB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const Frag4<NS> f = split4<NS>(f32x4{v[u][i].x, v[u][i].y, v[u][i].z, v[u][i].w});
#pragma unroll
        for (int n = 0; n < NS; ++n)
          *reinterpret_cast<bf16x4 *>(lds + n * img + tile * Lay<D>::ROWB + (4 * rb + i) * Lay<D>::RS + 8 * cb) = f.v[n];
      }
    }
  }
  template <int NS>
  __device__ __forceinline__ void store_tr(unsigned char *lds, int img, int tid) const {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int blk = tid + u * THREADS;
      if (kPartial && blk >= BLK) continue;
      const int tile = blk / (8 * CB), rb = (blk / CB) % 8, cb = blk % CB;
      unsigned char *base = lds + tile * Lay<D>::TRB + (4 * cb) * Lay<D>::TS + 8 * rb;
      const Frag4<NS> fx = split4<NS>(f32x4{v[u][0].x, v[u][1].x, v[u][2].x, v[u][3].x});
      const Frag4<NS> fy = split4<NS>(f32x4{v[u][0].y, v[u][1].y, v[u][2].y, v[u][3].y});
      const Frag4<NS> fz = split4<NS>(f32x4{v[u][0].z, v[u][1].z, v[u][2].z, v[u][3].z});
      const Frag4<NS> fw = split4<NS>(f32x4{v[u][0].w, v[u][1].w, v[u][2].w, v[u][3].w});
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        *reinterpret_cast<bf16x4 *>(base + n * img) = fx.v[n];
        *reinterpret_cast<bf16x4 *>(base + n * img + Lay<D>::TS) = fy.v[n];
        *reinterpret_cast<bf16x4 *>(base + n * img + 2 * Lay<D>::TS) = fz.v[n];
        *reinterpret_cast<bf16x4 *>(base + n * img + 3 * Lay<D>::TS) = fw.v[n];
      }
    }
  }
};

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int NS>
__device__ __forceinline__ f32x16 mfma_x(const Frag<NS> &a, const Frag<NS> &b, f32x16 c) {
  if constexpr (NS == 3) {  // smallest contributions first
    c = mfma_bf16(a.v[0], b.v[2], c);
    c = mfma_bf16(a.v[2], b.v[0], c);
    c = mfma_bf16(a.v[1], b.v[1], c);
    c = mfma_bf16(a.v[1], b.v[0], c);
    c = mfma_bf16(a.v[0], b.v[1], c);
  }
  return mfma_bf16(a.v[0], b.v[0], c);
}

// ---------------------------------------------------------------------------------------------- forward
// SPLIT = false: wave w owns queries (tile*4 + w)*32 .. +31 and walks all 4 key tiles of a stage.
// SPLIT = true:  the 4 waves share 32 queries, wave w takes key tile w of every stage; the partial
//                (m, l, O) are merged through LDS (decoder shapes: 256 / 512 queries).
// QT (SPLIT only): query tiles per workgroup -- wave (qt, ks) takes key tile ks of the stage for query tile qt, so every
//                staged K / V tile serves QT query tiles (attention.hip, mha_fwd_kernel: the split-key launches are
//                bound by the delivery of K / V from L2).
template <int D, bool SPLIT, bool GEN, int NW = 4, int NS = 1, int QT = 1>
__global__ __launch_bounds__(NW * QT * kWave, (D == 64 && NW == 4 && QT == 1 && (NS == 1 || !SPLIT) ? 2 : 1)) void mha_fwd_bf16_kernel(MhaParams p) {
  static_assert(QT == 1 || SPLIT, "several query tiles per workgroup: split-key form only");
  using L = Lay<D>;
  // SPLIT: one key tile per wave per stage; three-piece operands: two tiles per stage keep the long-sequence
  // kernel at 55 KB of LDS (two workgroups per CU)
  constexpr int NT = D / 32, KC = D / 16, TILES = (NS == 3 && !SPLIT) ? 2 : NW;
  static_assert(SPLIT || NW == 4, "the long-sequence kernel runs 4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IMG_K = TILES * L::ROWB, IMG_V = TILES * L::TRB;
  unsigned char *s_k = smem, *s_vt = smem + NS * IMG_K;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int qt = SPLIT ? w / NW : 0, ks = SPLIT ? w % NW : w;
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = SPLIT ? (th.tile * QT + qt) * kTile : (th.tile * 4 + w) * kTile;
  const int myq = q0 + l31;
  const bool wave_active = q0 < p.l;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

  Frag<NS> qf[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c)
    qf[c] = load_frag<NS>(qbase + static_cast<size_t>(myq < p.l ? myq : 0) * qstride + 16 * c + 8 * half, myq < p.l,
                           p.scale * kLog2e);  // S, running maximum and lse in log2 units (exp2 without a multiply)

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) o[t] = zero16();
  float m = -INFINITY, lsum = 0.f;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;

  Fetch<D, TILES, NW * QT * kWave> fk, fv;
  fk.load(kbase, kstride, 0, p.s, tid);
  fv.load(vbase, vstride, 0, p.s, tid);
  for (int sbase = 0; sbase < p.s; sbase += kTile * TILES) {
    __syncthreads();
    fk.template store_rm<NS>(s_k, IMG_K, tid);
    fv.template store_tr<NS>(s_vt, IMG_V, tid);
    __syncthreads();
    if (sbase + kTile * TILES < p.s) {
      fk.load(kbase, kstride, sbase + kTile * TILES, p.s, tid);
      fv.load(vbase, vstride, sbase + kTile * TILES, p.s, tid);
    }
    for (int tile = SPLIT ? ks : 0; tile < (SPLIT ? ks + 1 : TILES); ++tile) {
      const int s0 = sbase + tile * kTile;
      if (!wave_active || s0 >= p.s) break;
      const unsigned char *tk = s_k + tile * L::ROWB, *tv = s_vt + tile * L::TRB;

      f32x16 sacc = zero16();
#pragma unroll
      for (int c = 0; c < KC; ++c) sacc = mfma_x<NS>(read_rm<D, NS>(tk, IMG_K, l31, c, half), qf[c], sacc);
      // sacc[r] = scale * <q[myq], k[s0 + crow(r, half)]>
      float pr[16];
      float tmax = -INFINITY;
      if (!GEN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pr[r] = sacc[r];
          tmax = fmaxf(tmax, pr[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = s0 + crow(r, half);
          bool dead = key >= p.s;
          if (p.mask && !dead && myq < p.l) dead = p.mask[(static_cast<size_t>(bh) * p.l + myq) * p.s + key] != 0;
          pr[r] = dead ? -INFINITY : sacc[r];
          tmax = fmaxf(tmax, pr[r]);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m, tmax);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      if (__ballot(m_new != m) != 0ull) {  // lazy rescale (see attention.hip)
        const float alpha = fast_exp2(m - m_safe);
        lsum *= alpha;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        m = m_new;
      }
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = fast_exp2(pr[r] - m_safe);
        rs += pr[r];
      }
      lsum += rs;
      if (use_drop) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t hsh = drop_hash(dconst, myq, p.s, s0 + crow(r, half));
          pr[r] = drop_keep_lo(hsh, p.thresh16) ? pr[r] : 0.f;  // 1 / (1 - p): once, on the output row
          pr[r + 1] = drop_keep_hi(hsh, p.thresh16) ? pr[r + 1] : 0.f;
        }
      }
      // O^T[dv][q] += sum_key V[key][dv] P[q][key]:  A = V^T (transposed tile), B = P^T (registers)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const Frag<NS> pb = CODA_SPLIT8(NS, pr, jj);
#pragma unroll
        for (int t = 0; t < NT; ++t) o[t] = mfma_x<NS>(read_tr<D, NS>(tv, IMG_V, 32 * t + l31, jj, half), pb, o[t]);
      }
    }
  }

  lsum += __shfl_xor(lsum, 32);
  if (SPLIT) {  // merge the per-wave partial softmax states: [wave-1][NT*16 + 2][64 lanes] floats
    __syncthreads();
    float *s_f = reinterpret_cast<float *>(smem) + static_cast<size_t>(qt) * (NW - 1) * (NT * 16 + 2) * kWave;
    float *slot = s_f + static_cast<size_t>(ks > 0 ? ks - 1 : 0) * (NT * 16 + 2) * kWave;
    if (ks > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * kWave + lane] = o[t][r];
      slot[(NT * 16) * kWave + lane] = m;
      slot[(NT * 16 + 1) * kWave + lane] = lsum;
    }
    __syncthreads();
    if (ks > 0) return;
    float m_all = m;
    for (int ww = 1; ww < NW; ++ww) m_all = fmaxf(m_all, s_f[((ww - 1) * (NT * 16 + 2) + NT * 16) * kWave + lane]);
    const float m_ref = (m_all == -INFINITY) ? 0.f : m_all;
    const float f0 = fast_exp2(m - m_ref);
    lsum *= f0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= f0;
    for (int ww = 1; ww < NW; ++ww) {
      const float *sl = s_f + static_cast<size_t>(ww - 1) * (NT * 16 + 2) * kWave;
      const float fw = fast_exp2(sl[(NT * 16) * kWave + lane] - m_ref);
      lsum += sl[(NT * 16 + 1) * kWave + lane] * fw;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] += sl[(t * 16 + r) * kWave + lane] * fw;
    }
    m = m_all;
  }
  if (myq < p.l) {
    // o[t][r]: head-dim component 32t + crow(r, half) of query myq -> registers 4g..4g+3 are 4 consecutive floats
    const float inv = lsum > 0.f ? p.inv_keep / lsum : 0.f;
    float *orow = p.out + static_cast<size_t>(myq) * rstride + head_off;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(orow + 32 * t + 8 * g + 4 * half) =
            make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
    if (half == 0) p.lse[static_cast<size_t>(bh) * p.l + myq] = lsum > 0.f ? m * kLn2 + __logf(lsum) : -INFINITY;
  }
}

// ---------------------------------------------------------------------------------------------- dK / dV
// A wave owns 32 keys (K, V fragments in registers as B operands), query tiles come through LDS in both
// images (row-major: A operands of S = Q K^T and dP = dO V^T; transposed: B operands of dV = Pd^T dO and
// dK = dS^T Q).  QSPLIT as in attention.hip: the 4 waves share 32 keys and split the query tiles.
template <int D, bool QSPLIT, bool GEN, int NS = 1>
__global__ __launch_bounds__(kThreads, (D == 64 ? 2 : 1)) void mha_bwd_dkv_bf16_kernel(MhaBwdParams p) {
  using L = Lay<D>;
  // three-piece operands: one query tile per stage keeps the kernel at 55 KB of LDS, i.e. two workgroups per CU
  constexpr int NT = D / 32, KC = D / 16, QT = QSPLIT ? 4 : (NS == 3 ? 1 : 2);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IMG_R = QT * L::ROWB, IMG_T = QT * L::TRB;
  unsigned char *s_q = smem, *s_do = s_q + NS * IMG_R, *s_qt = s_do + NS * IMG_R, *s_dot = s_qt + NS * IMG_T;
  float *s_lse = reinterpret_cast<float *>(s_dot + NS * IMG_T), *s_delta = s_lse + QT * kTile;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int k0 = QSPLIT ? th.tile * kTile : (th.tile * 4 + w) * kTile;
  const int mykey = k0 + l31;
  const bool wave_active = k0 < p.s;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;
  const float *gbase = p.dout + head_off;

  Frag<NS> kf[KC], vf[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const size_t row = static_cast<size_t>(mykey < p.s ? mykey : 0);
    kf[c] = load_frag<NS>(kbase + row * kstride + 16 * c + 8 * half, mykey < p.s, 1.0f);
    vf[c] = load_frag<NS>(vbase + row * vstride + 16 * c + 8 * half, mykey < p.s, 1.0f);
  }
  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { dk[t] = zero16(); dv[t] = zero16(); }

  Fetch<D, QT> fq, fg;
  float r_lse = 0.f, r_delta = 0.f;
  auto fetch_rows = [&](int qb) {
    if (tid < kTile * QT) {
      const int qq = qb + tid;
      r_lse = qq < p.l ? p.lse[static_cast<size_t>(bh) * p.l + qq] * kLog2e : 0.f;  // log2 units
      r_delta = qq < p.l ? p.delta[static_cast<size_t>(bh) * p.l + qq] : 0.f;
    }
  };
  fq.load(qbase, qstride, 0, p.l, tid);
  fg.load(gbase, rstride, 0, p.l, tid);
  fetch_rows(0);
  for (int qb0 = 0; qb0 < p.l; qb0 += kTile * QT) {
    __syncthreads();
    fq.template store_rm<NS>(s_q, IMG_R, tid);
    fq.template store_tr<NS>(s_qt, IMG_T, tid);
    fg.template store_rm<NS>(s_do, IMG_R, tid);
    fg.template store_tr<NS>(s_dot, IMG_T, tid);
    if (tid < kTile * QT) { s_lse[tid] = r_lse; s_delta[tid] = r_delta; }
    __syncthreads();
    if (qb0 + kTile * QT < p.l) {
      fq.load(qbase, qstride, qb0 + kTile * QT, p.l, tid);
      fg.load(gbase, rstride, qb0 + kTile * QT, p.l, tid);
      fetch_rows(qb0 + kTile * QT);
    }
    if (!wave_active) continue;
    for (int qt = QSPLIT ? w : 0; qt < (QSPLIT ? w + 1 : QT); ++qt) {
      const int q0 = qb0 + qt * kTile;
      if (q0 >= p.l) break;
      const unsigned char *tq = s_q + qt * L::ROWB, *tdo = s_do + qt * L::ROWB;
      const unsigned char *tqt = s_qt + qt * L::TRB, *tdot = s_dot + qt * L::TRB;
      const float *t_lse = s_lse + qt * kTile, *t_delta = s_delta + qt * kTile;

      f32x16 sacc = zero16(), pacc = zero16();
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        sacc = mfma_x<NS>(read_rm<D, NS>(tq, IMG_R, l31, c, half), kf[c], sacc);
        pacc = mfma_x<NS>(read_rm<D, NS>(tdo, IMG_R, l31, c, half), vf[c], pacc);
      }
      // lane: key = mykey; register r: query q0 + crow(r, half)
      float pd[16], ds[16];
      const float sscale = p.scale * kLog2e;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = crow(r, half), qq = q0 + qi;
        const float lse = t_lse[qi];
        float prob;
        if (!GEN) {
          prob = fast_exp2(sacc[r] * sscale - lse);  // one fma
        } else {
          bool dead = qq >= p.l || mykey >= p.s;
          if (p.mask && !dead) dead = p.mask[(static_cast<size_t>(bh) * p.l + qq) * p.s + mykey] != 0;
          prob = (dead || lse == -INFINITY) ? 0.f : fast_exp2(sacc[r] * sscale - lse);
        }
        float keep = 1.f;
        if (use_drop) keep = drop_keep(drop_hash(dconst, qq, p.s, mykey), mykey, p.thresh16) ? p.inv_keep : 0.f;
        pd[r] = prob * keep;
        ds[r] = prob * (pacc[r] * keep - t_delta[qi]);  // * scale: once, on the dK rows
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const Frag<NS> pa = CODA_SPLIT8(NS, pd, jj), da = CODA_SPLIT8(NS, ds, jj);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          dv[t] = mfma_x<NS>(pa, read_tr<D, NS>(tdot, IMG_T, 32 * t + l31, jj, half), dv[t]);
          dk[t] = mfma_x<NS>(da, read_tr<D, NS>(tqt, IMG_T, 32 * t + l31, jj, half), dk[t]);
        }
      }
    }
  }

  if (QSPLIT) {  // sum the per-wave partial dK / dV: [wave-1][2*NT*16][64 lanes] floats
    __syncthreads();
    float *s_f = reinterpret_cast<float *>(smem);
    if (w > 0) {
      float *slot = s_f + static_cast<size_t>(w - 1) * (2 * NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          slot[(t * 16 + r) * kWave + lane] = dk[t][r];
          slot[((NT + t) * 16 + r) * kWave + lane] = dv[t][r];
        }
    }
    __syncthreads();
    if (w > 0) return;
    for (int ww = 1; ww < 4; ++ww) {
      const float *sl = s_f + static_cast<size_t>(ww - 1) * (2 * NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dk[t][r] += sl[(t * 16 + r) * kWave + lane];
          dv[t][r] += sl[((NT + t) * 16 + r) * kWave + lane];
        }
    }
  }
  // dk[t][r]: key k0 + crow(r, half), head-dim component 32t + l31 (one 128-B row segment per wave half)
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + crow(r, half);
      if (key < p.s) {
        float *dkrow = p.dk + (static_cast<size_t>(key) * p.b + bi) * p.lddk + hi * D + l31;
        float *dvrow = p.dv + (static_cast<size_t>(key) * p.b + bi) * p.lddv + hi * D + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          dkrow[32 * t] = dk[t][r] * p.scale;
          dvrow[32 * t] = dv[t][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- dQ
// A wave owns 32 queries (Q, dO fragments in registers as B operands of S^T = K Q^T and dP^T = V dO^T);
// K comes through LDS in both images (row-major for S^T, transposed for dQ = dS K), V row-major.
template <int D, bool SPLIT, bool GEN, int NW = 4, int NS = 1, int QT = 1>  // QT: see mha_fwd_bf16_kernel
__global__ __launch_bounds__(NW * QT * kWave, (D == 64 && NW == 4 && QT == 1 && (NS == 1 || !SPLIT) ? 2 : 1)) void mha_bwd_dq_bf16_kernel(MhaBwdParams p) {
  static_assert(QT == 1 || SPLIT, "several query tiles per workgroup: split-key form only");
  using L = Lay<D>;
  // three-piece operands: one key tile per stage (41 KB of LDS: three workgroups per CU by LDS, two by registers)
  constexpr int NT = D / 32, KC = D / 16, TILES = (NS == 3 && !SPLIT) ? 1 : NW;
  static_assert(SPLIT || NW == 4, "the long-sequence kernel runs 4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IMG_R = TILES * L::ROWB, IMG_T = TILES * L::TRB;
  unsigned char *s_k = smem, *s_v = s_k + NS * IMG_R, *s_kt = s_v + NS * IMG_R;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int qt = SPLIT ? w / NW : 0, ks = SPLIT ? w % NW : w;
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = SPLIT ? (th.tile * QT + qt) * kTile : (th.tile * 4 + w) * kTile;
  const int myq = q0 + l31;
  const bool wave_active = q0 < p.l;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

  Frag<NS> qf[KC], gf[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const size_t row = static_cast<size_t>(myq < p.l ? myq : 0);
    qf[c] = load_frag<NS>(qbase + row * qstride + 16 * c + 8 * half, myq < p.l, p.scale * kLog2e);
    gf[c] = load_frag<NS>(p.dout + row * rstride + head_off + 16 * c + 8 * half, myq < p.l, 1.0f);
  }
  float lse = 0.f, delta = 0.f;
  if (myq < p.l) {
    lse = p.lse[static_cast<size_t>(bh) * p.l + myq] * kLog2e;  // log2 units
    delta = p.delta[static_cast<size_t>(bh) * p.l + myq];
  }
  const float lse_eff = (myq < p.l && lse != -INFINITY) ? lse : INFINITY;  // exp2(x - inf) = 0 kills empty rows
  f32x16 dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dq[t] = zero16();

  Fetch<D, TILES, NW * QT * kWave> fk, fv;
  fk.load(kbase, kstride, 0, p.s, tid);
  fv.load(vbase, vstride, 0, p.s, tid);
  for (int sbase = 0; sbase < p.s; sbase += kTile * TILES) {
    __syncthreads();
    fk.template store_rm<NS>(s_k, IMG_R, tid);
    fk.template store_tr<NS>(s_kt, IMG_T, tid);
    fv.template store_rm<NS>(s_v, IMG_R, tid);
    __syncthreads();
    if (sbase + kTile * TILES < p.s) {
      fk.load(kbase, kstride, sbase + kTile * TILES, p.s, tid);
      fv.load(vbase, vstride, sbase + kTile * TILES, p.s, tid);
    }
    for (int tile = SPLIT ? ks : 0; tile < (SPLIT ? ks + 1 : TILES); ++tile) {
      const int s0 = sbase + tile * kTile;
      if (!wave_active || s0 >= p.s) break;
      const unsigned char *tk = s_k + tile * L::ROWB, *tv = s_v + tile * L::ROWB, *tkt = s_kt + tile * L::TRB;

      f32x16 sacc = zero16(), pacc = zero16();
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        sacc = mfma_x<NS>(read_rm<D, NS>(tk, IMG_R, l31, c, half), qf[c], sacc);
        pacc = mfma_x<NS>(read_rm<D, NS>(tv, IMG_R, l31, c, half), gf[c], pacc);
      }
      // lane: query = myq; register r: key s0 + crow(r, half)
      float ds[16];
      if (!GEN) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float keep0 = 1.f, keep1 = 1.f;
          if (use_drop) {
            const uint32_t hsh = drop_hash(dconst, myq, p.s, s0 + crow(r, half));
            keep0 = drop_keep_lo(hsh, p.thresh16) ? p.inv_keep : 0.f;
            keep1 = drop_keep_hi(hsh, p.thresh16) ? p.inv_keep : 0.f;
          }
          const float prob0 = fast_exp2(sacc[r] - lse_eff);
          const float prob1 = fast_exp2(sacc[r + 1] - lse_eff);
          ds[r] = prob0 * (pacc[r] * keep0 - delta);  // * scale: once, on the dQ rows
          ds[r + 1] = prob1 * (pacc[r + 1] * keep1 - delta);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = s0 + crow(r, half);
          bool dead = key >= p.s || myq >= p.l;
          if (p.mask && !dead) dead = p.mask[(static_cast<size_t>(bh) * p.l + myq) * p.s + key] != 0;
          const float prob = (dead || lse == -INFINITY) ? 0.f : fast_exp2(sacc[r] - lse);
          float keep = 1.f;
          if (use_drop) keep = drop_keep(drop_hash(dconst, myq, p.s, key), key, p.thresh16) ? p.inv_keep : 0.f;
          ds[r] = prob * (pacc[r] * keep - delta);
        }
      }
      // dQ[q][d] += sum_key dS[q][key] K[key][d]:  A = dS (registers), B = K (transposed tile)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const Frag<NS> da = CODA_SPLIT8(NS, ds, jj);
#pragma unroll
        for (int t = 0; t < NT; ++t) dq[t] = mfma_x<NS>(da, read_tr<D, NS>(tkt, IMG_T, 32 * t + l31, jj, half), dq[t]);
      }
    }
  }
  if (SPLIT) {  // sum the per-wave partial dQ: [wave-1][NT*16][64 lanes] floats
    __syncthreads();
    float *s_f = reinterpret_cast<float *>(smem) + static_cast<size_t>(qt) * (NW - 1) * (NT * 16) * kWave;
    if (ks > 0) {
      float *slot = s_f + static_cast<size_t>(ks - 1) * (NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * kWave + lane] = dq[t][r];
    }
    __syncthreads();
    if (ks > 0) return;
    for (int ww = 1; ww < NW; ++ww) {
      const float *sl = s_f + static_cast<size_t>(ww - 1) * (NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] += sl[(t * 16 + r) * kWave + lane];
    }
  }
  // dq[t][r]: query q0 + crow(r, half), head-dim component 32t + l31
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + crow(r, half);
      if (qq < p.l) {
        float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) row[32 * t] = dq[t][r] * p.scale;
      }
    }
  }
}

template <typename K>
int raise_lds(K kern, size_t bytes) {
  return raise_dynamic_lds(kern, bytes);  // per (kernel entry point, device), common.hip.h
}

// short query / key sequences (decoder: 256 or 512 object queries) use the split variants
// Query (key, for dK/dV) counts from which a wave owns its own 32 rows and the workgroup's waves share the staged
// tiles; below, the waves share 32 rows and split the other sequence.  CODA_ATTN_BF16_SPLIT_BELOW overrides (A/B).
inline int split_below() {
  static const int v = [] { const char *e = getenv("CODA_ATTN_BF16_SPLIT_BELOW"); return e ? atoi(e) : 1024; }();
  return v;
}
#define kSplitBelow split_below()

// Which problems the three-piece (NS = 3) kernels take; the others stay with the fp32-MFMA kernels of
// attention.hip (same fp32-level arithmetic either way, so the two can be mixed inside one backward).
// Measured on 2048 x 2048 (8 scenes x 4 heads, dropout 0.1; fp32-MFMA kernels: 351 / 646 / 465 us):
//   forward 270 us, dQ 376 us, dK/dV 786 us.  The six bf16 MFMAs per product are hidden, but splitting the
//   operands costs ~290 extra VALU instructions per 32-key tile (v_cvt_pk_bf16_f32, widen, subtract, twice) on top
//   of the soft-max's ~240, so the kernels are VALU-bound at 1.2-1.3x the fp32 kernels instead of the 2.5x the
//   MFMA arithmetic alone would give; dK/dV (two register-resident operand sets, two split accumulator images
//   per tile) spills and loses.  Hence: forward and long-sequence dQ only.
bool x3_takes_fwd(const MhaParams &p) { return p.l >= 1024; }
bool x3_takes_dkv(const MhaBwdParams &) { return false; }
bool x3_takes_dq(const MhaBwdParams &p) { return p.l >= 1024; }

template <int D, bool GEN, int NS>
int fwd_launch(const MhaParams &p, hipStream_t s) {
  using L = Lay<D>;
  int st;
  if (p.l >= kSplitBelow) {
    const size_t lds = static_cast<size_t>(NS == 3 ? 2 : 4) * NS * (L::ROWB + L::TRB);
    auto kern = mha_fwd_bf16_kernel<D, false, GEN, 4, NS>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile * 4), p.b * p.h), dim3(kThreads), lds, s, p);
  } else if (NS == 1 && D == 64 && p.s >= 1024 && split_query_tiles(ceil_div(p.l, 2 * kTile) * p.b * p.h) == 1) {
    // two query tiles per workgroup share the staged K / V tiles (half the L2 traffic), one 8-wave workgroup per CU
    // (the merge of the 2 x 3 partial soft-max states reuses the stage buffers and is the larger of the two)
    const size_t lds = std::max<size_t>(4 * (L::ROWB + L::TRB), sizeof(float) * 2 * 3 * (D / 32 * 16 + 2) * kWave);
    auto kern = mha_fwd_bf16_kernel<D, true, GEN, 4, 1, (NS == 1 && D == 64 ? 2 : 1)>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, 2 * kTile), p.b * p.h), dim3(8 * kWave), lds, s, p);
  } else if (NS == 1 && D == 64 && p.s >= 8 * kTile && ceil_div(p.l, kTile) * p.b * p.h <= 256) {
    // at most one workgroup per CU: 8 waves / 8 key tiles per stage put twice the bytes in flight per CU
    // (measured 64 -> 53 us at 256 x 2048, but 87 -> 101 us at 512 x 2048 where two 4-wave workgroups share a CU)
    const size_t lds = 8 * (L::ROWB + L::TRB);
    auto kern = mha_fwd_bf16_kernel<D, true, GEN, (D == 64 ? 8 : 4), 1>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(8 * kWave), lds, s, p);
  } else {
    const size_t lds = static_cast<size_t>(4) * NS * (L::ROWB + L::TRB);
    auto kern = mha_fwd_bf16_kernel<D, true, GEN, 4, NS>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(kThreads), lds, s, p);
  }
  return CODA_OK;
}

template <int D, bool GEN, int NS>
int dkv_launch(const MhaBwdParams &p, hipStream_t s) {
  using L = Lay<D>;
  int st;
  // (from 512 keys on: 512 x 512, the self-attention of a 512-query decoder, 60 -> 52 us)
  if (p.s >= kSplitBelow / 2) {
    constexpr int QT = NS == 3 ? 1 : 2;
    const size_t lds = static_cast<size_t>(QT) * NS * (2 * L::ROWB + 2 * L::TRB) + sizeof(float) * 2 * kTile * QT;
    auto kern = mha_bwd_dkv_bf16_kernel<D, false, GEN, NS>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.s, kTile * 4), p.b * p.h), dim3(kThreads), lds, s, p);
  } else {
    if (NS != 1) return CODA_EINVAL;  // x3_takes_dkv() keeps these shapes away
    const size_t lds = 4 * (2 * L::ROWB + 2 * L::TRB) + sizeof(float) * 2 * kTile * 4;
    auto kern = mha_bwd_dkv_bf16_kernel<D, true, GEN, 1>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.s, kTile), p.b * p.h), dim3(kThreads), lds, s, p);
  }
  return CODA_OK;
}

template <int D, bool GEN, int NS>
int dq_launch(const MhaBwdParams &p, hipStream_t s) {
  using L = Lay<D>;
  int st;
  if (p.l >= kSplitBelow) {
    const size_t lds = static_cast<size_t>(NS == 3 ? 1 : 4) * NS * (2 * L::ROWB + L::TRB);
    auto kern = mha_bwd_dq_bf16_kernel<D, false, GEN, 4, NS>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile * 4), p.b * p.h), dim3(kThreads), lds, s, p);
  } else if (NS != 1) {
    return CODA_EINVAL;  // x3_takes_dq() keeps these shapes away
  } else if (D == 64 && p.s >= 1024 && split_query_tiles(ceil_div(p.l, 2 * kTile) * p.b * p.h) == 1) {
    const size_t lds = std::max<size_t>(4 * (2 * L::ROWB + L::TRB), sizeof(float) * 2 * 3 * (D / 32 * 16) * kWave);  // (see fwd_launch)
    auto kern = mha_bwd_dq_bf16_kernel<D, true, GEN, 4, 1, (D == 64 ? 2 : 1)>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, 2 * kTile), p.b * p.h), dim3(8 * kWave), lds, s, p);
  } else if (D == 64 && p.s >= 8 * kTile && ceil_div(p.l, kTile) * p.b * p.h <= 256) {
    const size_t lds = 8 * (2 * L::ROWB + L::TRB);
    auto kern = mha_bwd_dq_bf16_kernel<D, true, GEN, (D == 64 ? 8 : 4), 1>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(8 * kWave), lds, s, p);
  } else {
    const size_t lds = 4 * (2 * L::ROWB + L::TRB);
    auto kern = mha_bwd_dq_bf16_kernel<D, true, GEN, 4, 1>;
    if ((st = raise_lds(kern, lds)) != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(kThreads), lds, s, p);
  }
  return CODA_OK;
}

template <typename P>
bool general(const P &p) { return p.mask != nullptr || (p.l % kTile) != 0 || (p.s % kTile) != 0; }

template <int NS>
int fwd_any(const MhaParams &p, int d, hipStream_t s) {
  const bool gen = general(p);
  if (d == 64) return gen ? fwd_launch<64, true, NS>(p, s) : fwd_launch<64, false, NS>(p, s);
  return gen ? fwd_launch<128, true, NS>(p, s) : fwd_launch<128, false, NS>(p, s);
}
template <int NS>
int dkv_any(const MhaBwdParams &p, int d, hipStream_t s) {
  const bool gen = general(p);
  if (d == 64) return gen ? dkv_launch<64, true, NS>(p, s) : dkv_launch<64, false, NS>(p, s);
  return gen ? dkv_launch<128, true, NS>(p, s) : dkv_launch<128, false, NS>(p, s);
}
template <int NS>
int dq_any(const MhaBwdParams &p, int d, hipStream_t s) {
  const bool gen = general(p);
  if (d == 64) return gen ? dq_launch<64, true, NS>(p, s) : dq_launch<64, false, NS>(p, s);
  return gen ? dq_launch<128, true, NS>(p, s) : dq_launch<128, false, NS>(p, s);
}

}  // namespace

int mha_fwd_bf16(const MhaParams &p, int d, hipStream_t s) { return fwd_any<1>(p, d, s); }
int mha_bwd_dkv_bf16(const MhaBwdParams &p, int d, hipStream_t s) { return dkv_any<1>(p, d, s); }
int mha_bwd_dq_bf16(const MhaBwdParams &p, int d, hipStream_t s) { return dq_any<1>(p, d, s); }

// three-piece operands (fp32-level results on the bf16 matrix cores); the *_takes predicates say which
// problems have a kernel, the caller keeps the fp32-MFMA kernel for the rest
bool mha_x3_takes_fwd(const MhaParams &p, int d) { return d == 64 && x3_takes_fwd(p); }
bool mha_x3_takes_dkv(const MhaBwdParams &p, int d) { return d == 64 && x3_takes_dkv(p); }
bool mha_x3_takes_dq(const MhaBwdParams &p, int d) { return d == 64 && x3_takes_dq(p); }
int mha_fwd_x3(const MhaParams &p, int d, hipStream_t s) { return fwd_any<3>(p, d, s); }
int mha_bwd_dkv_x3(const MhaBwdParams &p, int d, hipStream_t s) { return dkv_any<3>(p, d, s); }
int mha_bwd_dq_x3(const MhaBwdParams &p, int d, hipStream_t s) { return dq_any<3>(p, d, s); }

}  // namespace coda
