// attention.hip -- fused multi-head attention core (forward + backward), fp32 MFMA.
//
// Replaces the baddbmm / softmax / dropout / bmm sequence inside
// torch.nn.MultiheadAttention that the reference's encoder / decoder layers call
// (models/transformer.py:470-471,566-573): the (B*h, L, S) fp32 probability
// tensor (67 MB per scene per encoder layer) is never materialised.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31],
// C/D col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)):
//  * Both GEMMs of the forward are evaluated TRANSPOSED so that a lane always owns
//    one query column: S^T = K Q^T (A = K rows, B = Q rows) puts 16 keys of one
//    query in the 16 accumulator registers of a lane -> the online-softmax row
//    reductions are in-register plus ONE cross-half shuffle; O^T = V^T P^T takes
//    those same registers unchanged as its B operand and keeps O with lane = query,
//    so the rescale by exp(m_old - m_new) is a per-lane scalar multiply.
//  * The two k-slots of a k-step are mapped to head-dim components (kk, D/2 + kk):
//    a lane's A/B fragment is D/2 CONSECUTIVE floats of one row -> ds_read_b128 /
//    global float4 loads, LDS rows padded by 16 B (conflict-free b128 reads).
//  * In O^T = V^T P^T the output row i <-> head-dim component is free; choosing
//    dv = (D/32)*i + tile lets one ds_read_b64/b128 feed all D/32 output tiles.
// The backward runs as two kernels (dK/dV owned per key block, dQ owned per query
// block): no atomics, deterministic, at the price of recomputing S twice.
#include "coda_attention.h"
#include "attention_common.hip.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace coda {
PendingTiming &pending_timing() {
  thread_local PendingTiming t;
  return t;
}
namespace {

// Cooperative copy of `rows` x D floats (row stride `gstride` floats) into a padded LDS tile.
template <int D, int THREADS, int ROWS = kTile>
__device__ __forceinline__ void load_tile(float *lds, const float *g, size_t gstride, int row0,
                                          int nrows_total, int tid) {
  constexpr int LS = D + 4;
  for (int i = tid; i < ROWS * D / 4; i += THREADS) {
    const int row = i / (D / 4), c4 = i % (D / 4);
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows_total)
      val = *reinterpret_cast<const float4 *>(g + static_cast<size_t>(row0 + row) * gstride + c4 * 4);
    *reinterpret_cast<float4 *>(lds + row * LS + c4 * 4) = val;
  }
}

// The same copy in two halves, for double buffering: global -> registers (issued before the
// compute on the current LDS buffer), registers -> the other LDS buffer (after it).
template <int D, int THREADS, int ROWS>
__device__ __forceinline__ void fetch_tile(float4 (&regs)[ROWS * D / 4 / THREADS], const float *g, size_t gstride,
                                           int row0, int nrows_total, int tid) {
#pragma unroll
  for (int j = 0; j < ROWS * D / 4 / THREADS; ++j) {
    const int i = tid + j * THREADS;
    const int row = i / (D / 4), c4 = i % (D / 4);
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows_total)
      val = *reinterpret_cast<const float4 *>(g + static_cast<size_t>(row0 + row) * gstride + c4 * 4);
    regs[j] = val;
  }
}
template <int D, int THREADS, int ROWS>
__device__ __forceinline__ void store_tile(float *lds, const float4 (&regs)[ROWS * D / 4 / THREADS], int tid) {
  constexpr int LS = D + 4;
#pragma unroll
  for (int j = 0; j < ROWS * D / 4 / THREADS; ++j) {
    const int i = tid + j * THREADS;
    const int row = i / (D / 4), c4 = i % (D / 4);
    *reinterpret_cast<float4 *>(lds + row * LS + c4 * 4) = regs[j];
  }
}

// SPLIT = false: every wave owns its own block of 32 queries and all waves share one K/V tile
//                 per step (long query sequences: the encoder).
// SPLIT = true:  all QW waves work on the SAME 32 queries and split the keys (QW tiles are
//                 staged per step, wave w takes tile w); the partial (m, l, O) are merged
//                 through LDS at the end.  This is what fills the chip for the decoder, whose
//                 256 queries give only 8 query blocks per (scene, head).
// DB (non-SPLIT only): the same LDS holds two half-size stages; the global loads of stage i+1 are
//                 in flight while stage i is computed and land in the other buffer afterwards --
//                 one barrier per stage and no wave parked on an HBM round trip.
// QT (SPLIT only): query tiles per workgroup.  The split-key launches of the decoder are bound by the delivery of K / V
//                 from L2 (8 workgroups per head each read the head's whole K and V: 256 MB per launch at 4.7 TB/s), not
//                 by the MFMAs; with QT = 2 a workgroup is 2 x QW waves, wave (qt, ks) takes key tile ks of the stage for
//                 query tile qt, and every staged tile serves two query tiles -- half the traffic.
template <int D, int QW, bool SPLIT, bool GEN, bool DB = false, int QT = 1>
__global__ __launch_bounds__(QW * QT * kWave) void mha_fwd_kernel(MhaParams p) {
  static_assert(QT == 1 || SPLIT, "several query tiles per workgroup: split-key form only");
  constexpr int HD = D / 2, NT = D / 32, LS = D + 4, THREADS = QW * QT * kWave;
  // K/V tiles staged per step (SPLIT: one per wave; else all waves walk all of them)
  // (DB: two stages in LDS -- half-size ones in the same footprint for the long-sequence kernel,
  //  full-size ones, i.e. twice the LDS, for the split-key kernel where every wave needs its tile)
  constexpr int TILES = (DB && !SPLIT) ? QW / 2 : QW;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *s_k = s_dyn;
  float *s_v = s_dyn + TILES * kTile * LS;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int qt = SPLIT ? w / QW : 0, ks = SPLIT ? w % QW : w;  // SPLIT: query tile of the workgroup, key split
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = SPLIT ? (th.tile * QT + qt) * kTile : (th.tile * QW + w) * kTile;
  const int my_tile = SPLIT ? ks : 0;
  const int myq = q0 + l31;
  const bool wave_active = q0 < p.l;  // wave-uniform
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;  // dense outputs
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

  float qf[HD];
  const float qscale = p.scale * kLog2e;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (myq < p.l)
      t = *reinterpret_cast<const float4 *>(qbase + static_cast<size_t>(myq) * qstride + half * HD + c);
    // scale * log2(e) folded into Q: S, the running maximum and lse are kept in log2 units (exp2 without a multiply)
    qf[c] = t.x * qscale; qf[c + 1] = t.y * qscale; qf[c + 2] = t.z * qscale; qf[c + 3] = t.w * qscale;
  }

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;

  constexpr int kStageFloats = 2 * TILES * kTile * LS;  // one K + V stage
  constexpr int NLD = DB ? TILES * kTile * D / 4 / THREADS : 1;
  constexpr int FROWS = DB ? TILES * kTile : THREADS * 4 / D;
  float4 rk[NLD], rv[NLD];
  if (DB) {  // prologue: stage 0 straight into buffer 0
    fetch_tile<D, THREADS, FROWS>(rk, kbase, kstride, 0, p.s, tid);
    fetch_tile<D, THREADS, FROWS>(rv, vbase, vstride, 0, p.s, tid);
    store_tile<D, THREADS, FROWS>(s_k, rk, tid);
    store_tile<D, THREADS, FROWS>(s_v, rv, tid);
    __syncthreads();
  }
  int stage = 0;
  for (int sbase = 0; sbase < p.s; sbase += kTile * TILES, ++stage) {
    const bool more = DB && sbase + kTile * TILES < p.s;
    if (!DB) {
      __syncthreads();
      load_tile<D, THREADS, kTile * TILES>(s_k, kbase, kstride, sbase, p.s, tid);
      load_tile<D, THREADS, kTile * TILES>(s_v, vbase, vstride, sbase, p.s, tid);
      __syncthreads();
    } else {
      s_k = s_dyn + (stage & 1) * kStageFloats;
      s_v = s_k + TILES * kTile * LS;
      if (more) {  // next stage: loads in flight during this stage's MFMAs
        fetch_tile<D, THREADS, FROWS>(rk, kbase, kstride, sbase + kTile * TILES, p.s, tid);
        fetch_tile<D, THREADS, FROWS>(rv, vbase, vstride, sbase + kTile * TILES, p.s, tid);
      }
    }
    for (int tile = SPLIT ? my_tile : 0; tile < (SPLIT ? my_tile + 1 : TILES); ++tile) {
    const int s0 = sbase + tile * kTile;
    if (!wave_active || s0 >= p.s) break;
    const float *tk = s_k + tile * kTile * LS, *tv = s_v + tile * kTile * LS;

    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 kf = *reinterpret_cast<const float4 *>(tk + l31 * LS + half * HD + c);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c + 1], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c + 2], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c + 3], sacc, 0, 0, 0);
    }
    // sacc[r] = scale * <q[myq], k[s0 + crow(r, half)]>
    float pr[16];
    float tmax = -INFINITY;
    // GEN = false (host-checked): no mask, L and S multiples of the tile -> nothing to mask out
    constexpr bool plain = !GEN;
    if (plain) {
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
        if (p.mask && !dead && myq < p.l)
          dead = p.mask[(static_cast<size_t>(bh) * p.l + myq) * p.s + key] != 0;
        pr[r] = dead ? -INFINITY : sacc[r];
        tmax = fmaxf(tmax, pr[r]);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    // Lazy rescale with slack: the reference point m of the exponentials only moves when some row's maximum has
    // outgrown it by more than kMaxSlack (log2 units) -- the probabilities of a tile are then at most 2^kMaxSlack
    // instead of 1, which changes nothing in out = o / lsum and lse = m + log lsum, and the accumulators (a rescale
    // is a read-modify-write of all of them: 160 of the ~730 non-MFMA instructions of a tile) are left alone after
    // the first tiles instead of in the ~30 % of the tiles in which no row at all changes its maximum.
    if (__ballot(m_new > m + kMaxSlack) != 0ull) {
      const float m_to = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m - m_to);
      lsum *= alpha;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m = m_new;
    }
    const float m_safe = (m == -INFINITY) ? 0.f : m;
    f32x2 rs2 = {0.f, 0.f};  // pairs: the subtraction and the row sum as packed fp32 operations
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 arg = f32x2{pr[r], pr[r + 1]} - f32x2{m_safe, m_safe};
      pr[r] = fast_exp2(arg[0]);
      pr[r + 1] = fast_exp2(arg[1]);
      rs2 += f32x2{pr[r], pr[r + 1]};
    }
    lsum += rs2[0] + rs2[1];
    if (use_drop) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {  // keys crow(r), crow(r)+1: one aligned pair
        const uint32_t hsh = drop_hash(dconst, myq, p.s, s0 + crow(r, half));
        pr[r] = drop_keep_lo(hsh, p.thresh16) ? pr[r] : 0.f;  // 1 / (1 - p) is applied once, to the output row
        pr[r + 1] = drop_keep_hi(hsh, p.thresh16) ? pr[r + 1] : 0.f;
      }
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *vrow = tv + crow(r, half) * LS + NT * l31;
      if (NT == 2) {
        const float2 vv = *reinterpret_cast<const float2 *>(vrow);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, pr[r], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, pr[r], o[1], 0, 0, 0);
      } else {
        const float4 vv = *reinterpret_cast<const float4 *>(vrow);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, pr[r], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, pr[r], o[1], 0, 0, 0);
        o[2 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, pr[r], o[2 % NT], 0, 0, 0);
        o[3 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, pr[r], o[3 % NT], 0, 0, 0);
      }
    }
    }  // tile
    if (DB) {
      if (more) {
        float *nk = s_dyn + ((stage + 1) & 1) * kStageFloats;
        store_tile<D, THREADS, FROWS>(nk, rk, tid);
        store_tile<D, THREADS, FROWS>(nk + TILES * kTile * LS, rv, tid);
      }
      __syncthreads();
    }
  }

  lsum += __shfl_xor(lsum, 32);
  if (SPLIT && QW > 1) {
    // merge the per-wave partial softmax states: slot layout [wave-1][NT*16 + 2][64 lanes]
    __syncthreads();  // everyone is done with the K/V tiles
    float *sq = s_dyn + static_cast<size_t>(qt) * (QW - 1) * (NT * 16 + 2) * kWave;  // this query tile's slots
    float *slot = sq + static_cast<size_t>(ks > 0 ? ks - 1 : 0) * (NT * 16 + 2) * kWave;
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
    for (int ww = 1; ww < QW; ++ww) m_all = fmaxf(m_all, sq[((ww - 1) * (NT * 16 + 2) + NT * 16) * kWave + lane]);
    const float m_ref = (m_all == -INFINITY) ? 0.f : m_all;
    const float f0 = fast_exp2(m - m_ref);
    lsum *= f0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= f0;
    for (int ww = 1; ww < QW; ++ww) {
      const float *sl = sq + static_cast<size_t>(ww - 1) * (NT * 16 + 2) * kWave;
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
    const float inv = lsum > 0.f ? p.inv_keep / lsum : 0.f;
    float *orow = p.out + static_cast<size_t>(myq) * rstride + head_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dv = NT * crow(r, half);
      if (NT == 2) {
        *reinterpret_cast<float2 *>(orow + dv) = make_float2(o[0][r] * inv, o[1][r] * inv);
      } else {
        *reinterpret_cast<float4 *>(orow + dv) =
            make_float4(o[0][r] * inv, o[1][r] * inv, o[2 % NT][r] * inv, o[3 % NT][r] * inv);
      }
    }
    if (half == 0)
      p.lse[static_cast<size_t>(bh) * p.l + myq] = lsum > 0.f ? m * kLn2 + __logf(lsum) : -INFINITY;
  }
}

// Split-key forward WITHOUT LDS staging (round 3).  In the split-key decomposition every K / V tile is used by exactly
// one wave, so staging it through LDS buys no reuse -- it only costs the wave 32 VMEM issues + 32 ds_write_b128 per
// tile, a barrier per stage whose skew nothing hides (one wave per SIMD), and the LDS round trip in front of the
// MFMAs.  Here the fragments go from L2 straight into registers in the layout the MFMAs consume:
//   K (A operand of S^T = K Q^T):   lane (key = l31, half) reads the D/2 consecutive floats of its key row
//                                    -> D/8 float4 loads, each key row's two cache lines consumed completely;
//   V (A operand of O^T = V^T P^T): for accumulator row r the lanes of a half read key row crow(r, half) at
//                                    columns NT*l31.. -> 16 loads of NT floats, 256 B contiguous per half-wave.
// Two register sets: the loads of a wave's NEXT tile are issued before the MFMAs of the current one (a tile is
// ~2.5 us of work: any L2 / HBM latency is covered), no barrier anywhere in the key loop.  LDS is used for the
// final merge of the QW partial soft-max states only (26 KB), so the waves of a workgroup never wait for each other.
template <int D, int QW, bool GEN>
__global__ __launch_bounds__(QW * kWave, (QW >= 8 && D == 64) ? 2 : 1) void mha_fwd_direct_kernel(MhaParams p) {
  constexpr int HD = D / 2, NT = D / 32;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = th.tile * kTile;
  const int myq = q0 + l31;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

  float qf[HD];
  const float qscale = p.scale * kLog2e;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (myq < p.l) t = *reinterpret_cast<const float4 *>(qbase + static_cast<size_t>(myq) * qstride + half * HD + c);
    qf[c] = t.x * qscale; qf[c + 1] = t.y * qscale; qf[c + 2] = t.z * qscale; qf[c + 3] = t.w * qscale;
  }
  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;

  struct Frag {
    float4 k[HD / 4];
    float v[16][NT];
  };
  auto load = [&](Frag &f, int s0) {
    const int krow = s0 + l31;
    const bool kin = !GEN || krow < p.s;
    const float *kp = kbase + static_cast<size_t>(kin ? krow : 0) * kstride + half * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const float4 t = *reinterpret_cast<const float4 *>(kp + 4 * c);
      f.k[c] = kin ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int vrow = s0 + crow(r, half);
      const bool vin = !GEN || vrow < p.s;
      const float *vp = vbase + static_cast<size_t>(vin ? vrow : 0) * vstride + NT * l31;
      if (NT == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(vp);
        f.v[r][0] = vin ? t.x : 0.f; f.v[r][1] = vin ? t.y : 0.f;
      } else {
        const float4 t = *reinterpret_cast<const float4 *>(vp);
        f.v[r][0] = vin ? t.x : 0.f; f.v[r][1] = vin ? t.y : 0.f; f.v[r][2 % NT] = vin ? t.z : 0.f; f.v[r][3 % NT] = vin ? t.w : 0.f;
      }
    }
  };
  auto compute = [&](const Frag &f, int s0) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 kf = f.k[c / 4];
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c + 1], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c + 2], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c + 3], sacc, 0, 0, 0);
    }
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
    if (__ballot(m_new > m + kMaxSlack) != 0ull) {  // lazy rescale with slack (see mha_fwd_kernel)
      const float m_to = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m - m_to);
      lsum *= alpha;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m = m_new;
    }
    const float m_safe = (m == -INFINITY) ? 0.f : m;
    f32x2 rs2 = {0.f, 0.f};  // pairs: the subtraction and the row sum as packed fp32 operations
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 arg = f32x2{pr[r], pr[r + 1]} - f32x2{m_safe, m_safe};
      pr[r] = fast_exp2(arg[0]);
      pr[r + 1] = fast_exp2(arg[1]);
      rs2 += f32x2{pr[r], pr[r + 1]};
    }
    lsum += rs2[0] + rs2[1];
    if (use_drop) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t hsh = drop_hash(dconst, myq, p.s, s0 + crow(r, half));
        pr[r] = drop_keep_lo(hsh, p.thresh16) ? pr[r] : 0.f;
        pr[r + 1] = drop_keep_hi(hsh, p.thresh16) ? pr[r + 1] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.v[r][t], pr[r], o[t], 0, 0, 0);
    }
  };

  const int ntile = (p.s + kTile - 1) / kTile;
  if (q0 < p.l) {  // wave-uniform (all waves of a workgroup share the query tile)
    // The prefetch of the wave's next tile is UNCONDITIONAL (the last one re-reads the final tile): behind an `if` the
    // compiler has to assume at the join that the loads were NOT issued, and the waits it then places in front of the
    // current tile's MFMAs (vmcnt(15) .. vmcnt(0), counted for 24 outstanding loads) drain the prefetch as well whenever
    // it was -- every tile waited for the next tile's K / V before its second GEMM.  (Worth 1-2 % here, 6 % in the
    // head-width-128 dQ kernel: with one wave per SIMD the exposed soft-max arithmetic is the larger share.)
    Frag fa, fb;
    int t = w;
    const int last = ntile - 1;
    if (t < ntile) {
      load(fa, t * kTile);
      while (true) {
        load(fb, min(t + QW, last) * kTile);
        compute(fa, t * kTile);
        t += QW;
        if (t >= ntile) break;
        load(fa, min(t + QW, last) * kTile);
        compute(fb, t * kTile);
        t += QW;
        if (t >= ntile) break;
      }
    }
  }

  lsum += __shfl_xor(lsum, 32);
  if (QW > 1) {  // merge the per-wave partial soft-max states: slot layout [wave-1][NT*16 + 2][64 lanes]
    float *slot = s_dyn + static_cast<size_t>(w > 0 ? w - 1 : 0) * (NT * 16 + 2) * kWave;
    if (w > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * kWave + lane] = o[t][r];
      slot[(NT * 16) * kWave + lane] = m;
      slot[(NT * 16 + 1) * kWave + lane] = lsum;
    }
    __syncthreads();
    if (w > 0) return;
    float m_all = m;
    for (int ww = 1; ww < QW; ++ww) m_all = fmaxf(m_all, s_dyn[((ww - 1) * (NT * 16 + 2) + NT * 16) * kWave + lane]);
    const float m_ref = (m_all == -INFINITY) ? 0.f : m_all;
    const float f0 = fast_exp2(m - m_ref);
    lsum *= f0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= f0;
    for (int ww = 1; ww < QW; ++ww) {
      const float *sl = s_dyn + static_cast<size_t>(ww - 1) * (NT * 16 + 2) * kWave;
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
    const float inv = lsum > 0.f ? p.inv_keep / lsum : 0.f;
    float *orow = p.out + static_cast<size_t>(myq) * rstride + head_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dv = NT * crow(r, half);
      if (NT == 2) {
        *reinterpret_cast<float2 *>(orow + dv) = make_float2(o[0][r] * inv, o[1][r] * inv);
      } else {
        *reinterpret_cast<float4 *>(orow + dv) =
            make_float4(o[0][r] * inv, o[1][r] * inv, o[2 % NT][r] * inv, o[3 % NT][r] * inv);
      }
    }
    if (half == 0) p.lse[static_cast<size_t>(bh) * p.l + myq] = lsum > 0.f ? m * kLn2 + __logf(lsum) : -INFINITY;
  }
}

template <int D>
__global__ __launch_bounds__(256) void mha_delta_kernel(MhaBwdParams p) {
  // one thread per (q, b, h) row
  const size_t row = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  const size_t nrows = static_cast<size_t>(p.l) * p.b * p.h;
  if (row >= nrows) return;
  const float *o = p.out + row * D, *g = p.dout + row * D;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    const float4 a = *reinterpret_cast<const float4 *>(o + c);
    const float4 d = *reinterpret_cast<const float4 *>(g + c);
    acc += a.x * d.x + a.y * d.y + a.z * d.z + a.w * d.w;
  }
  const int hh = row % p.h, bb = (row / p.h) % p.b, qq = row / (static_cast<size_t>(p.h) * p.b);
  p.delta[(static_cast<size_t>(bb) * p.h + hh) * p.l + qq] = acc;
}

// dK / dV: a wave owns 32 keys (K, V fragments in registers), loops over query tiles.
// S = Q K^T is evaluated UN-transposed here (A = Q rows, B = K rows): a lane then owns one
// key column and 16 queries in registers, which is the A-operand layout of dV = Pd^T dO and
// dK = dS^T Q (k-dimension = query).
// QSPLIT = false: the KW waves of a block own KW different key tiles and walk all queries
//                 (long key sequences: the encoder, the decoder's memory).
// QSPLIT = true:  all KW waves own the SAME 32 keys and split the query tiles (KW tiles are
//                 staged per step, wave w takes tile w); the partial dK / dV are summed
//                 through LDS at the end.  Short key sequences (the decoder's 256 queries
//                 attending to themselves) would otherwise leave one wave per CU.
// DB (non-QSPLIT): two single-tile stages of Q / dO in the same LDS, double buffered like the
//                 forward kernel's K / V staging.
// (A variant with the V fragments of a lane in LDS instead of 32 registers removed the kernel's 72 B / lane of scratch
// and was not faster -- 648 vs 626 us -- so the spill is not what holds this kernel; removed in round 4.)
// WDS: dS (before the soft-max scale, exactly what this kernel feeds into dK) is also written to p.ds (B*H, L, S), so
//      that dQ = scale * dS K is one plain GEMM (mha_bwd_dq_gemm_kernel) instead of a second kernel that recomputes S
//      and dP: the backward executes 10 instead of 14 units of L * S * d flops per head.
template <int D, int KW, bool QSPLIT, bool GEN, bool DB = false, bool WDS = false>
__global__ __launch_bounds__(KW * kWave, ((D == 64 && !QSPLIT) ? 2 : 1)) void mha_bwd_dkv_kernel(MhaBwdParams p) {
  constexpr int HD = D / 2, NT = D / 32, LS = D + 4, THREADS = KW * kWave;
  constexpr int QT = QSPLIT ? KW : (DB ? 1 : 2);  // query tiles staged per step
  static_assert(!(DB && QSPLIT), "double buffering is implemented for the long-key-sequence kernel");
  constexpr int kStageFloats = 2 * QT * kTile * LS + 2 * QT * kTile;  // Q, dO tiles + lse, delta rows
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *s_q = s_dyn;
  float *s_do = s_q + QT * kTile * LS;
  float *s_lse = s_do + QT * kTile * LS;
  float *s_delta = s_lse + QT * kTile;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int k0 = QSPLIT ? th.tile * kTile : (th.tile * KW + w) * kTile;
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

  float kf[HD], vf[HD];  // B operands: K[mykey][half*HD + c], V[mykey][half*HD + c]
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a;
    if (mykey < p.s) {
      a = *reinterpret_cast<const float4 *>(kbase + static_cast<size_t>(mykey) * kstride + half * HD + c);
      b4 = *reinterpret_cast<const float4 *>(vbase + static_cast<size_t>(mykey) * vstride + half * HD + c);
    }
    kf[c] = a.x; kf[c + 1] = a.y; kf[c + 2] = a.z; kf[c + 3] = a.w;
    vf[c] = b4.x; vf[c + 1] = b4.y; vf[c + 2] = b4.z; vf[c + 3] = b4.w;
  }
  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  constexpr int NLD = DB ? QT * kTile * D / 4 / THREADS : 1;
  constexpr int FROWS = DB ? QT * kTile : THREADS * 4 / D;
  float4 rq[NLD], rg[NLD];
  float r_lse = 0.f, r_delta = 0.f;
  auto fetch_rows = [&](int qb) {  // lse / delta of the stage's queries: one value per thread
    if (tid < kTile * QT) {
      const int qq = qb + tid;
      r_lse = qq < p.l ? p.lse[static_cast<size_t>(bh) * p.l + qq] * kLog2e : 0.f;  // log2 units
      r_delta = qq < p.l ? p.delta[static_cast<size_t>(bh) * p.l + qq] : 0.f;
    }
  };
  if (DB) {  // prologue: stage 0 straight into buffer 0
    fetch_tile<D, THREADS, FROWS>(rq, qbase, qstride, 0, p.l, tid);
    fetch_tile<D, THREADS, FROWS>(rg, p.dout + head_off, rstride, 0, p.l, tid);
    fetch_rows(0);
    store_tile<D, THREADS, FROWS>(s_q, rq, tid);
    store_tile<D, THREADS, FROWS>(s_do, rg, tid);
    if (tid < kTile * QT) { s_lse[tid] = r_lse; s_delta[tid] = r_delta; }
    if constexpr (WDS) lds_only_barrier(); else __syncthreads();
  }
  int stage = 0;
  for (int qbase0 = 0; qbase0 < p.l; qbase0 += kTile * QT, ++stage) {
    const bool more = DB && qbase0 + kTile * QT < p.l;
    if (!DB) {
      if constexpr (WDS) lds_only_barrier(); else __syncthreads();
      load_tile<D, THREADS, kTile * QT>(s_q, qbase, qstride, qbase0, p.l, tid);
      load_tile<D, THREADS, kTile * QT>(s_do, p.dout + head_off, rstride, qbase0, p.l, tid);
      if (tid < kTile * QT) {
        const int qq = qbase0 + tid;
        s_lse[tid] = qq < p.l ? p.lse[static_cast<size_t>(bh) * p.l + qq] * kLog2e : 0.f;
        s_delta[tid] = qq < p.l ? p.delta[static_cast<size_t>(bh) * p.l + qq] : 0.f;
      }
      if constexpr (WDS) lds_only_barrier(); else __syncthreads();
    } else {
      s_q = s_dyn + (stage & 1) * kStageFloats;
      s_do = s_q + QT * kTile * LS;
      s_lse = s_do + QT * kTile * LS;
      s_delta = s_lse + QT * kTile;
      if (more) {  // next stage: loads in flight during this stage's MFMAs
        fetch_tile<D, THREADS, FROWS>(rq, qbase, qstride, qbase0 + kTile * QT, p.l, tid);
        fetch_tile<D, THREADS, FROWS>(rg, p.dout + head_off, rstride, qbase0 + kTile * QT, p.l, tid);
        fetch_rows(qbase0 + kTile * QT);
      }
    }
    if (wave_active) {
    for (int qt = QSPLIT ? w : 0; qt < (QSPLIT ? w + 1 : QT); ++qt) {
    const int q0 = qbase0 + qt * kTile;
    if (q0 >= p.l) break;
    const float *tq = s_q + qt * kTile * LS, *tdo = s_do + qt * kTile * LS;
    const float *t_lse = s_lse + qt * kTile, *t_delta = s_delta + qt * kTile;

    // S[q][key] and dP[q][key]: A = Q / dO rows (lane = query), B = K / V rows (lane = key)
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 qa = *reinterpret_cast<const float4 *>(tq + l31 * LS + half * HD + c);
      const float4 ga = *reinterpret_cast<const float4 *>(tdo + l31 * LS + half * HD + c);
      const float4 vv = make_float4(vf[c], vf[c + 1], vf[c + 2], vf[c + 3]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kf[c], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, vv.x, pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kf[c + 1], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, vv.y, pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kf[c + 2], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, vv.z, pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kf[c + 3], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, vv.w, pacc, 0, 0, 0);
    }
    // lane: key = mykey, register r: query q0 + crow(r, half)
    float pd[16], ds[16];
    const float sscale = p.scale * kLog2e;
    constexpr bool plain = !GEN;  // no mask, L and S multiples of the tile
    if (plain) {
      // One hash word decides a PAIR of adjacent keys (drop_hash ignores the key's low bit), and here adjacent keys are
      // adjacent lanes: lane parity `par` computes the word of query crow(r) + par of every register pair (r, r + 1)
      // and gets the other one from its neighbour with a DPP swap -- 8 hashes per tile and lane instead of 16, the
      // same words as before.
      const int par = l31 & 1;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float keep0 = 1.f, keep1 = 1.f;
        if (use_drop) {
          const uint32_t mine = drop_hash(dconst, q0 + crow(r, half) + par, p.s, mykey);
          const uint32_t other = __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
          keep0 = drop_keep(par ? other : mine, mykey, p.thresh16) ? p.inv_keep : 0.f;
          keep1 = drop_keep(par ? mine : other, mykey, p.thresh16) ? p.inv_keep : 0.f;
        }
        // the pair's arithmetic as packed fp32 operations (v_pk_fma_f32 / v_pk_mul_f32: two elements per instruction)
        const int qi = crow(r, half);
        const f32x2 lse2 = {t_lse[qi], t_lse[qi + 1]}, del2 = {t_delta[qi], t_delta[qi + 1]};
        const f32x2 keep2 = {keep0, keep1}, s2 = {sacc[r], sacc[r + 1]}, dp2 = {pacc[r], pacc[r + 1]};
        const f32x2 arg = __builtin_elementwise_fma(s2, f32x2{sscale, sscale}, -lse2);  // lse = -inf cannot occur here
        const f32x2 prob = {fast_exp2(arg[0]), fast_exp2(arg[1])};
        const f32x2 pdv = prob * keep2;
        const f32x2 dsv = prob * __builtin_elementwise_fma(dp2, keep2, -del2);  // * scale: once, on the dK rows
        pd[r] = pdv[0]; pd[r + 1] = pdv[1];
        ds[r] = dsv[0]; ds[r + 1] = dsv[1];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = crow(r, half), qq = q0 + qi;
        bool dead = qq >= p.l || mykey >= p.s;
        if (p.mask && !dead) dead = p.mask[(static_cast<size_t>(bh) * p.l + qq) * p.s + mykey] != 0;
        const float lse = t_lse[qi];
        float prob = (dead || lse == -INFINITY) ? 0.f : fast_exp2(__fmaf_rn(sacc[r], sscale, -lse));
        float keep = 1.f;
        if (use_drop) keep = drop_keep(drop_hash(dconst, qq, p.s, mykey), mykey, p.thresh16) ? p.inv_keep : 0.f;
        pd[r] = prob * keep;
        ds[r] = prob * __fmaf_rn(pacc[r], keep, -t_delta[qi]);  // * scale: once, on the dK rows
      }
    }
    if (WDS) {  // lane = key: the 32 lanes of a half-wave write 128 contiguous bytes of one query's row
      float *dsb = p.ds + (static_cast<size_t>(bh) * p.l + q0) * p.s + k0;  // wave-uniform base, 32-bit lane offsets
      const uint32_t off0 = static_cast<uint32_t>(4 * half) * static_cast<uint32_t>(p.s) + static_cast<uint32_t>(l31);
#pragma unroll
      for (int r = 0; r < 16; ++r)  // streaming stores: 16.8 MB per head must not push Q / dO out of the XCD's L2
        __builtin_nontemporal_store(ds[r], dsb + off0 + static_cast<uint32_t>((r & 3) + 8 * (r >> 2)) * static_cast<uint32_t>(p.s));
    }
    // dV^T? no: dV[key][dv] += sum_q Pd[q][key] dO[q][dv]  (A = Pd^T: lane = key, k = query)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = crow(r, half);
      const float *grow = tdo + qi * LS + NT * l31;
      const float *qrow = tq + qi * LS + NT * l31;
      if (NT == 2) {
        const float2 g2 = *reinterpret_cast<const float2 *>(grow);
        const float2 q2 = *reinterpret_cast<const float2 *>(qrow);
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.x, dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.y, dv[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.x, dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.y, dk[1], 0, 0, 0);
      } else {
        const float4 g4 = *reinterpret_cast<const float4 *>(grow);
        const float4 q4 = *reinterpret_cast<const float4 *>(qrow);
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g4.x, dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g4.y, dv[1], 0, 0, 0);
        dv[2 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g4.z, dv[2 % NT], 0, 0, 0);
        dv[3 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g4.w, dv[3 % NT], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q4.x, dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q4.y, dk[1], 0, 0, 0);
        dk[2 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q4.z, dk[2 % NT], 0, 0, 0);
        dk[3 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q4.w, dk[3 % NT], 0, 0, 0);
      }
    }
    }  // query tile
    }  // wave_active
    if (DB) {
      if (more) {
        float *nq = s_dyn + ((stage + 1) & 1) * kStageFloats;
        store_tile<D, THREADS, FROWS>(nq, rq, tid);
        store_tile<D, THREADS, FROWS>(nq + QT * kTile * LS, rg, tid);
        if (tid < kTile * QT) {
          nq[2 * QT * kTile * LS + tid] = r_lse;
          nq[2 * QT * kTile * LS + QT * kTile + tid] = r_delta;
        }
      }
      if constexpr (WDS) lds_only_barrier(); else __syncthreads();
    }
  }

  if (QSPLIT && KW > 1) {  // sum the per-wave partial dK / dV: [wave-1][2*NT*16][64 lanes]
    if constexpr (WDS) lds_only_barrier(); else __syncthreads();
    if (w > 0) {
      float *slot = s_dyn + static_cast<size_t>(w - 1) * (2 * NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          slot[(t * 16 + r) * kWave + lane] = dk[t][r];
          slot[((NT + t) * 16 + r) * kWave + lane] = dv[t][r];
        }
    }
    if constexpr (WDS) lds_only_barrier(); else __syncthreads();
    if (w > 0) return;
    for (int ww = 1; ww < KW; ++ww) {
      const float *sl = s_dyn + static_cast<size_t>(ww - 1) * (2 * NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dk[t][r] += sl[(t * 16 + r) * kWave + lane];
          dv[t][r] += sl[((NT + t) * 16 + r) * kWave + lane];
        }
    }
  }
  // dk[t][r]: row i = crow(r, half) = key within the tile, column j = l31 <-> component NT*l31 + t
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + crow(r, half);
      if (key < p.s) {
        float *dkrow = p.dk + (static_cast<size_t>(key) * p.b + bi) * p.lddk + hi * D + NT * l31;
        float *dvrow = p.dv + (static_cast<size_t>(key) * p.b + bi) * p.lddv + hi * D + NT * l31;
        if (NT == 2) {
          *reinterpret_cast<float2 *>(dkrow) = make_float2(dk[0][r] * p.scale, dk[1][r] * p.scale);
          *reinterpret_cast<float2 *>(dvrow) = make_float2(dv[0][r], dv[1][r]);
        } else {
          *reinterpret_cast<float4 *>(dkrow) = make_float4(dk[0][r] * p.scale, dk[1][r] * p.scale, dk[2 % NT][r] * p.scale,
                                                             dk[3 % NT][r] * p.scale);
          *reinterpret_cast<float4 *>(dvrow) = make_float4(dv[0][r], dv[1][r], dv[2 % NT][r], dv[3 % NT][r]);
        }
      }
    }
  }
}

// ---- dK / dV with S and dP on the bf16 matrix cores at fp32 accuracy (round 6) ------------------------------------
// Of the 8 units of L * S * d this kernel executes per head, the two that only FEED the soft-max backward -- S = Q K^T
// (recomputed) and dP = dO V^T -- have both operands in memory: tiles that are staged once per workgroup and stage
// (Q, dO) or once per wave (K, V).  Those operands are split into three bf16 pieces (x = hi + mid + lo exactly, as in
// gemm_x3.hip) where they are staged, and S / dP become six piece products each on v_mfma_f32_32x32x16_bf16, fp32
// accumulation: 24 MFMAs of 32 cycles per 32 x 32 tile instead of 32 MFMAs of 64 cycles -- 1536 instead of 4096 matrix
// cycles per stage for the pair.  dV = Pd^T dO and dK = dS^T Q stay on the fp32 MFMA: their A operands are the
// probabilities themselves, one split per ELEMENT, which is what holds the all-bf16x3 backward of attention_bf16.hip.
//  * K / V pieces of the wave's 32 keys: hi and mid in 2 x 32 registers per lane (as many as the fp32 fragments they
//    replace), the lo pieces -- one product in six -- in LDS, each lane reading back what it wrote (with all three in
//    registers the kernel spilled 54 of them).
//  * Q / dO pieces: [piece][query][64 bf16], rows 144 B apart (conflict-free ds_read_b128), written by the threads that
//    stage the fp32 tile (16 elements of each per thread and stage); the fp32 tiles stay, as dV / dK's B operands.
//  * The bf16 matrix core's accumulate truncates toward -inf (gemm_x3.hip, tools/x3_bias.py): a relative 1e-9 at k = 64,
//    but of ONE sign.  The staged Q / dO pieces carry the sign (-1)^query, so S and dP arrive as (-1)^query * value -- a
//    register's query parity is the parity of its index, the sign folds into constants of the soft-max backward -- and
//    the drift alternates from one query row to the next: sums over queries (dK, dV) and over tokens (every weight
//    gradient behind dQ) see noise, not a drift.
//  * One stage = one query tile; everything in LDS is single-buffered (45 KB), the next tile waits in registers:
//    barrier -- store (split) -- barrier per stage, the second workgroup of the CU fills the SIMDs meanwhile.
// Plain problems only (no mask, whole tiles), head width 64; WDS as in mha_bwd_dkv_kernel.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Bf3x4 {
  bf16x4 p[3];
};
// x = hi + mid + lo exactly (round-to-nearest-even conversions; both residuals are exact in fp32)
__device__ __forceinline__ Bf3x4 split3(f32x4 x) {
  Bf3x4 r;
  r.p[0] = __builtin_convertvector(x, bf16x4);
  x = x - __builtin_convertvector(r.p[0], f32x4);
  r.p[1] = __builtin_convertvector(x, bf16x4);
  x = x - __builtin_convertvector(r.p[1], f32x4);
  r.p[2] = __builtin_convertvector(x, bf16x4);
  return r;
}

constexpr int kPieceRow = 144;                 // bytes: 64 bf16 + 16
constexpr int kPiecePlane = kTile * kPieceRow;  // one piece of one 32 x 64 tile
constexpr size_t kDkvX3Lds = sizeof(float) * (2 * kTile * 68 + 2 * kTile) + 6 * kPiecePlane + 2 * 4 * 4096;  // + K / V lo pieces of 4 waves: 76.25 KB, two workgroups per CU

template <int KW, bool WDS>
__global__ __launch_bounds__(KW * kWave, 2) void mha_bwd_dkv_x3_kernel(MhaBwdParams p) {
  constexpr int D = 64, NT = 2, LS = D + 4, THREADS = KW * kWave;
  constexpr int NLD = kTile * D / 4 / THREADS;  // float4 per thread and tile
  static_assert(NLD >= 1 && kTile * D / 4 % THREADS == 0, "tile must divide over the threads");
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *s_q = s_dyn;
  float *s_do = s_q + kTile * LS;
  float *s_lse = s_do + kTile * LS;
  float *s_delta = s_lse + kTile;
  unsigned char *s_qp = reinterpret_cast<unsigned char *>(s_delta + kTile);  // three planes
  unsigned char *s_gp = s_qp + 3 * kPiecePlane;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const int aoff = l31 * kPieceRow + 16 * half;  // bytes: this lane's fragment of k block 0 in a piece plane
  unsigned char *s_klo = s_gp + 3 * kPiecePlane + w * 8192 + 16 * lane;  // this wave's K / V lo pieces [k block][lane], this lane's
  unsigned char *s_vlo = s_klo + 4096;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int k0 = (th.tile * KW + w) * kTile;
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

  // B operands of S and dP: the pieces of K[mykey][16 j + 8 half + (0..7)], V likewise, j = 16-k block
  bf16x8 kp[2][4], vp[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    if (mykey < p.s) {
      const float *kr = kbase + static_cast<size_t>(mykey) * kstride + 16 * j + 8 * half;
      const float *vr = vbase + static_cast<size_t>(mykey) * vstride + 16 * j + 8 * half;
      a0 = *reinterpret_cast<const f32x4 *>(kr); a1 = *reinterpret_cast<const f32x4 *>(kr + 4);
      b0 = *reinterpret_cast<const f32x4 *>(vr); b1 = *reinterpret_cast<const f32x4 *>(vr + 4);
    }
    const Bf3x4 ka = split3(a0), kb = split3(a1), va = split3(b0), vb = split3(b1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      kp[q][j] = __builtin_shufflevector(ka.p[q], kb.p[q], 0, 1, 2, 3, 4, 5, 6, 7);
      vp[q][j] = __builtin_shufflevector(va.p[q], vb.p[q], 0, 1, 2, 3, 4, 5, 6, 7);
    }
    *reinterpret_cast<bf16x8 *>(s_klo + 1024 * j) = __builtin_shufflevector(ka.p[2], kb.p[2], 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8 *>(s_vlo + 1024 * j) = __builtin_shufflevector(va.p[2], vb.p[2], 0, 1, 2, 3, 4, 5, 6, 7);
  }
  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  float4 rq[NLD], rg[NLD];
  float r_lse = 0.f, r_delta = 0.f;
  auto fetch = [&](int qb) {
    fetch_tile<D, THREADS, kTile>(rq, qbase, qstride, qb, p.l, tid);
    fetch_tile<D, THREADS, kTile>(rg, p.dout + head_off, rstride, qb, p.l, tid);
    if (tid < kTile) {
      const int qq = qb + tid;
      r_lse = qq < p.l ? p.lse[static_cast<size_t>(bh) * p.l + qq] * kLog2e : 0.f;  // log2 units
      r_delta = qq < p.l ? p.delta[static_cast<size_t>(bh) * p.l + qq] : 0.f;
    }
  };
  auto store = [&]() {
    store_tile<D, THREADS, kTile>(s_q, rq, tid);
    store_tile<D, THREADS, kTile>(s_do, rg, tid);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = tid + j * THREADS;
      const int row = i / (D / 4), c4 = i % (D / 4);
      const float sg = (row & 1) ? -1.f : 1.f;
      const Bf3x4 pq = split3(f32x4{rq[j].x * sg, rq[j].y * sg, rq[j].z * sg, rq[j].w * sg});
      const Bf3x4 pg = split3(f32x4{rg[j].x * sg, rg[j].y * sg, rg[j].z * sg, rg[j].w * sg});
      const int off = row * kPieceRow + c4 * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        *reinterpret_cast<bf16x4 *>(s_qp + q * kPiecePlane + off) = pq.p[q];
        *reinterpret_cast<bf16x4 *>(s_gp + q * kPiecePlane + off) = pg.p[q];
      }
    }
    if (tid < kTile) { s_lse[tid] = r_lse; s_delta[tid] = r_delta; }
  };
  fetch(0);
  store();
  lds_only_barrier();

  const float sscale = p.scale * kLog2e;
  for (int q0 = 0; q0 < p.l; q0 += kTile) {
    const bool more = q0 + kTile < p.l;
    if (more) fetch(q0 + kTile);  // in flight during this stage
    if (wave_active) {
      // (-1)^query * S[q][key] and (-1)^query * dP[q][key]: A = Q / dO piece rows (lane = query), B = K / V pieces (lane = key)
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 aq[3], ag[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          aq[q] = *reinterpret_cast<const bf16x8 *>(s_qp + q * kPiecePlane + aoff + 32 * j);
          ag[q] = *reinterpret_cast<const bf16x8 *>(s_gp + q * kPiecePlane + aoff + 32 * j);
        }
        // the six products of order <= 2, small ones first
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[1], kp[1][j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[1], vp[1][j], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], *reinterpret_cast<const bf16x8 *>(s_klo + 1024 * j), sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[0], *reinterpret_cast<const bf16x8 *>(s_vlo + 1024 * j), pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[2], kp[0][j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[2], vp[0][j], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], kp[1][j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[0], vp[1][j], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[1], kp[0][j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[1], vp[0][j], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], kp[0][j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[0], vp[0][j], pacc, 0, 0, 0);
      }
      // lane: key = mykey, register r: query q0 + crow(r, half), whose parity is r & 1 (the staged sign)
      float pd[16], ds[16];
      const int par = l31 & 1;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float keep0 = 1.f, keep1 = 1.f;
        if (use_drop) {  // one hash word per pair of adjacent keys (see mha_bwd_dkv_kernel)
          const uint32_t mine = drop_hash(dconst, q0 + crow(r, half) + par, p.s, mykey);
          const uint32_t other = __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
          keep0 = drop_keep(par ? other : mine, mykey, p.thresh16) ? p.inv_keep : 0.f;
          keep1 = drop_keep(par ? mine : other, mykey, p.thresh16) ? p.inv_keep : 0.f;
        }
        const int qi = crow(r, half);
        const f32x2 lse2 = {s_lse[qi], s_lse[qi + 1]}, del2 = {s_delta[qi], s_delta[qi + 1]};
        const f32x2 keep2 = {keep0, keep1}, keeps2 = {keep0, -keep1};
        const f32x2 s2 = {sacc[r], sacc[r + 1]}, dp2 = {pacc[r], pacc[r + 1]};
        const f32x2 arg = __builtin_elementwise_fma(s2, f32x2{sscale, -sscale}, -lse2);
        const f32x2 prob = {fast_exp2(arg[0]), fast_exp2(arg[1])};
        const f32x2 pdv = prob * keep2;
        const f32x2 dsv = prob * __builtin_elementwise_fma(dp2, keeps2, -del2);  // * scale: once, on the dK rows
        pd[r] = pdv[0]; pd[r + 1] = pdv[1];
        ds[r] = dsv[0]; ds[r + 1] = dsv[1];
      }
      if (WDS) {  // lane = key: the 32 lanes of a half-wave write 128 contiguous bytes of one query's row
        float *dsb = p.ds + (static_cast<size_t>(bh) * p.l + q0) * p.s + k0;  // wave-uniform base, 32-bit lane offsets
        const uint32_t off0 = static_cast<uint32_t>(4 * half) * static_cast<uint32_t>(p.s) + static_cast<uint32_t>(l31);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_nontemporal_store(ds[r], dsb + off0 + static_cast<uint32_t>((r & 3) + 8 * (r >> 2)) * static_cast<uint32_t>(p.s));
      }
      // dV[key][c] += sum_q Pd[q][key] dO[q][c], dK[key][c] += sum_q dS[q][key] Q[q][c]  (A: lane = key, k = query)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = crow(r, half);
        const float2 g2 = *reinterpret_cast<const float2 *>(s_do + qi * LS + NT * l31);
        const float2 q2 = *reinterpret_cast<const float2 *>(s_q + qi * LS + NT * l31);
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.x, dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.y, dv[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.x, dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.y, dk[1], 0, 0, 0);
      }
    }
    lds_only_barrier();  // every wave is done with this stage's tiles
    if (more) store();
    lds_only_barrier();
  }
  // dk[t][r]: row i = crow(r, half) = key within the tile, column j = l31 <-> component NT*l31 + t
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + crow(r, half);
      if (key < p.s) {
        float *dkrow = p.dk + (static_cast<size_t>(key) * p.b + bi) * p.lddk + hi * D + NT * l31;
        float *dvrow = p.dv + (static_cast<size_t>(key) * p.b + bi) * p.lddv + hi * D + NT * l31;
        *reinterpret_cast<float2 *>(dkrow) = make_float2(dk[0][r] * p.scale, dk[1][r] * p.scale);
        *reinterpret_cast<float2 *>(dvrow) = make_float2(dv[0][r], dv[1][r]);
      }
    }
  }
}

// ---- the whole backward of a short query sequence against a long key sequence in ONE kernel (round 6) -------------
// The decoder's cross-attention (256 queries x 2048 keys): the two-kernel form recomputes S in both kernels and dP in
// the second one, 14 units of L * S * d per head for 8 algorithmic.  Here the dK/dV kernel's own dS also yields dQ:
// 10 units.  A workgroup = KW waves x 32 keys walks the query tiles like mha_bwd_dkv_kernel (S un-transposed: lane =
// key, registers = queries, the A-operand layout of dV and dK).  dQ = dS K contracts over KEYS, which sit in lanes:
//  * every wave writes its 32 x 32 dS tile TRANSPOSED into one shared LDS tile [query][KW * 32 keys];
//  * after a barrier the workgroup's dQ tile (32 queries x 64 components over its KW * 32 keys) is formed with
//    v_mfma_f32_16x16x4_f32: wave w owns components 16 w .. 16 w + 15 of both 16-query halves (8 accumulator
//    registers), A = dS^T rows and B = K^T rows as ds_read_b128 (K^T of the workgroup's keys: written once, [64][keys]);
//    nothing is summed across waves;
//  * the tile goes to a partial-sum workspace [head][key block][query tile][2048] in accumulator order (every store
//    instruction one contiguous 256 bytes); mha_dq_reduce_kernel sums the key blocks in FIXED order, scales and writes
//    dQ: deterministic, no atomics, no waiting among workgroups.
// One Q / dO tile in LDS (the next one is in flight in registers during the stage's second half; delta = rowsum(dO * O)
// is formed on its way into LDS), two LDS-only barriers per stage: 68 KB, two workgroups per CU.  Plain problems only (no mask, L % 32 == 0, S % (32 KW) == 0, head width 64).

// -DCODA_ATTN_PROF (tools/attn_phase_probe.py): shader-clock stamps of wave 0 of every workgroup of the short one-kernel
// backward, written over `delta` (which that kernel does not use): [workgroup][16] clocks since the kernel's first stamp.
#ifdef CODA_ATTN_PROF
#define CODA_PROF_STAMP(i)                                                                                   \
  do {                                                                                                       \
    __builtin_amdgcn_s_waitcnt(0);                                                                           \
    const long long t_now = clock64();                                                                       \
    if ((i) == 0) prof_t0 = t_now;                                                                           \
    if (threadIdx.x == 0)                                                                                    \
      p.delta[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = static_cast<float>(t_now - prof_t0); \
  } while (0)
#else
#define CODA_PROF_STAMP(i)
#endif

template <int KW>
__global__ __launch_bounds__(KW * kWave, 2) void mha_bwd_fused_kernel(MhaBwdParams p) {
  constexpr int D = 64, HD = 32, NT = 2, LS = D + 4, THREADS = KW * kWave, KB = KW * kTile, TS = KB + 4;
  constexpr int NLD = kTile * D / 4 / THREADS;
  static_assert(KW == 4, "component split of the dQ tile: four waves x 16 components");
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *s_q = s_dyn;                // [32][68]
  float *s_do = s_q + kTile * LS;    // [32][68]
  float *s_lse = s_do + kTile * LS;  // [32]
  float *s_delta = s_lse + kTile;    // [32]
  float *s_kt = s_delta + kTile;     // [64][KB + 4]: K^T of the workgroup's keys
  float *s_ds = s_kt + D * TS;       // [32][KB + 4]: dS^T of the stage

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int k0 = (th.tile * KW + w) * kTile;
  const int mykey = k0 + l31;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

  float vf[HD];  // B operand of dP: V[mykey][half*HD + c]
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 a = *reinterpret_cast<const float4 *>(kbase + static_cast<size_t>(mykey) * kstride + half * HD + c);
    const float4 b4 = *reinterpret_cast<const float4 *>(vbase + static_cast<size_t>(mykey) * vstride + half * HD + c);
    float *kt = s_kt + (half * HD + c) * TS + w * kTile + l31;  // K^T: B operand of S (this wave's keys) and of dQ (all)
    kt[0] = a.x; kt[TS] = a.y; kt[2 * TS] = a.z; kt[3 * TS] = a.w;
    vf[c] = b4.x; vf[c + 1] = b4.y; vf[c + 2] = b4.z; vf[c + 3] = b4.w;
  }
  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  // Q / dO / O rows of the next stage in registers; delta[row] = sum(dO * O) is formed when they go to LDS (the 16
  // threads that copy a row each hold four products): no launch for it
  float4 rq[NLD], rg[NLD], ro[NLD];
  float r_lse = 0.f;
  // (thread: rows tid / 16 and tid / 16 + 16 of the tile -- a uniform base + the thread's own 32-bit offset)
  const uint32_t qoff = static_cast<uint32_t>(tid / (D / 4)) * static_cast<uint32_t>(qstride) + (tid % (D / 4)) * 4;
  const uint32_t roff = static_cast<uint32_t>(tid / (D / 4)) * static_cast<uint32_t>(rstride) + (tid % (D / 4)) * 4;
  auto fetch_stage = [&](int qb) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int row0 = qb + j * (THREADS / (D / 4));
      rq[j] = *reinterpret_cast<const float4 *>(qbase + static_cast<size_t>(row0) * qstride + qoff);
      rg[j] = *reinterpret_cast<const float4 *>(p.dout + head_off + static_cast<size_t>(row0) * rstride + roff);
      ro[j] = *reinterpret_cast<const float4 *>(p.out + head_off + static_cast<size_t>(row0) * rstride + roff);
    }
    if (tid < kTile) r_lse = p.lse[static_cast<size_t>(bh) * p.l + qb + tid] * kLog2e;  // log2 units
  };
  auto store_stage = [&]() {
    store_tile<D, THREADS, kTile>(s_q, rq, tid);
    store_tile<D, THREADS, kTile>(s_do, rg, tid);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      float part_d = rg[j].x * ro[j].x + rg[j].y * ro[j].y + rg[j].z * ro[j].z + rg[j].w * ro[j].w;
      part_d += __shfl_xor(part_d, 1, kWave);
      part_d += __shfl_xor(part_d, 2, kWave);
      part_d += __shfl_xor(part_d, 4, kWave);
      part_d += __shfl_xor(part_d, 8, kWave);
      if (tid % (D / 4) == 0) s_delta[(tid + j * THREADS) / (D / 4)] = part_d;
    }
    if (tid < kTile) s_lse[tid] = r_lse;
  };
  fetch_stage(0);
  store_stage();
  lds_only_barrier();

  const int nqt = p.l / kTile;
  const int l15 = lane & 15, g = lane >> 4;
  // this workgroup's partial dQ tiles: [bh][key block][query tile][(w, query half, register)][lane]
  float *part = p.ds + (static_cast<size_t>(bh) * gridDim.x + th.tile) * nqt * (kTile * D) + (w * 8) * kWave + lane;
  const float *a_row = s_ds + l15 * TS + g * 32, *b_row = s_kt + (16 * w + l15) * TS + g * 32;
  const float sscale = p.scale * kLog2e;
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * kTile;
    const bool more = qt + 1 < nqt;
    // S[q][key] and dP[q][key]: A = Q / dO rows (lane = query), B = K / V rows (lane = key)
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 qa = *reinterpret_cast<const float4 *>(s_q + l31 * LS + half * HD + c);
      const float4 ga = *reinterpret_cast<const float4 *>(s_do + l31 * LS + half * HD + c);
      // K's fragments come back from the K^T tile (conflict-free ds_read_b32) instead of 32 registers held for the whole
      // kernel: no spills at two workgroups per CU, 98 instead of 102 us on 256 x 2048 x 8 scenes
      const float *kt = s_kt + (half * HD + c) * TS + w * kTile + l31;
      const float k0v = kt[0], k1v = kt[TS], k2v = kt[2 * TS], k3v = kt[3 * TS];
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, k0v, sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, vf[c], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, k1v, sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, vf[c + 1], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, k2v, sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, vf[c + 2], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, k3v, sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, vf[c + 3], pacc, 0, 0, 0);
    }
    // lane: key = mykey, register r: query q0 + crow(r, half); the soft-max backward of mha_bwd_dkv_kernel's plain path
    float pd[16], ds[16];
    {
      const int par = l31 & 1;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float keep0 = 1.f, keep1 = 1.f;
        if (use_drop) {
          const uint32_t mine = drop_hash(dconst, q0 + crow(r, half) + par, p.s, mykey);
          const uint32_t other = __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
          keep0 = drop_keep(par ? other : mine, mykey, p.thresh16) ? p.inv_keep : 0.f;
          keep1 = drop_keep(par ? mine : other, mykey, p.thresh16) ? p.inv_keep : 0.f;
        }
        const int qi = crow(r, half);
        const f32x2 lse2 = {s_lse[qi], s_lse[qi + 1]}, del2 = {s_delta[qi], s_delta[qi + 1]};
        const f32x2 keep2 = {keep0, keep1}, s2 = {sacc[r], sacc[r + 1]}, dp2 = {pacc[r], pacc[r + 1]};
        const f32x2 arg = __builtin_elementwise_fma(s2, f32x2{sscale, sscale}, -lse2);
        const f32x2 prob = {fast_exp2(arg[0]), fast_exp2(arg[1])};
        const f32x2 pdv = prob * keep2;
        const f32x2 dsv = prob * __builtin_elementwise_fma(dp2, keep2, -del2);  // * scale: on the dK rows / in the dQ sum
        pd[r] = pdv[0]; pd[r + 1] = pdv[1];
        ds[r] = dsv[0]; ds[r + 1] = dsv[1];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_ds[crow(r, half) * TS + w * kTile + l31] = ds[r];
    // next stage: its loads are in flight during the 64 MFMAs below (issued here, not at the top of the stage: the
    // soft-max above is where the registers are short)
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch_stage(q0 + kTile);
    // dV[key][dv] += sum_q Pd[q][key] dO[q][dv],  dK[key][c] += sum_q dS[q][key] Q[q][c]  (A: lane = key, k = query)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = crow(r, half);
      const float2 g2 = *reinterpret_cast<const float2 *>(s_do + qi * LS + NT * l31);
      const float2 q2 = *reinterpret_cast<const float2 *>(s_q + qi * LS + NT * l31);
      dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.x, dv[0], 0, 0, 0);
      dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.y, dv[1], 0, 0, 0);
      dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.x, dk[0], 0, 0, 0);
      dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.y, dk[1], 0, 0, 0);
    }
    lds_only_barrier();  // dS^T complete; every wave is done with the Q / dO tile
    if (more) store_stage();
    // dQ tile: rows = queries (two halves of 16), columns = components 16 w + l15, contraction over the KB keys;
    // k-slot g of step (j, e) <-> key 32 g + 4 j + e
    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KB / 16; ++j) {
      const float4 a0 = *reinterpret_cast<const float4 *>(a_row + 4 * j);
      const float4 a1 = *reinterpret_cast<const float4 *>(a_row + 16 * TS + 4 * j);
      const float4 bb = *reinterpret_cast<const float4 *>(b_row + 4 * j);
      dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bb.x, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bb.x, dq1, 0, 0, 0);
      dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bb.y, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bb.y, dq1, 0, 0, 0);
      dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bb.z, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bb.z, dq1, 0, 0, 0);
      dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bb.w, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bb.w, dq1, 0, 0, 0);
    }
    {
      float *pt = part + static_cast<size_t>(qt) * (kTile * D);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pt[r * kWave] = dq0[r];
        pt[(4 + r) * kWave] = dq1[r];
      }
    }
    lds_only_barrier();  // next Q / dO tile visible; dS^T free again
  }

  // dk[t][r]: row i = crow(r, half) = key within the tile, column j = l31 <-> component NT*l31 + t
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = k0 + crow(r, half);
    float *dkrow = p.dk + (static_cast<size_t>(key) * p.b + bi) * p.lddk + hi * D + NT * l31;
    float *dvrow = p.dv + (static_cast<size_t>(key) * p.b + bi) * p.lddv + hi * D + NT * l31;
    *reinterpret_cast<float2 *>(dkrow) = make_float2(dk[0][r] * p.scale, dk[1][r] * p.scale);
    *reinterpret_cast<float2 *>(dvrow) = make_float2(dv[0][r], dv[1][r]);
  }
}

// The same for SHORT key sequences (the decoder's 256 queries attending to themselves): a workgroup owns ONE key tile and
// its KW waves split the query tiles like mha_bwd_dkv_kernel<QSPLIT> (wave w takes tile w of the KW staged per step), so
// a wave's dQ tile over the workgroup's 32 keys is nobody else's: dS goes transposed into the wave's own (spent) Q tile
// in LDS, dQ = dS K is 32 v_mfma_f32_32x32x2_f32 (A = dS^T rows as ds_read_b128, B = the key tile's rows from LDS), the
// partial tiles [head][key tile][query tile][2048] are summed by mha_dq_reduce_kernel<1>.  delta = rowsum(dO * O) is
// formed while the Q / dO tiles are staged (the 16 threads that copy a row each hold four products): no launch for it.
// The per-wave partial dK / dV are summed through LDS at the end as in the two-kernel form.
template <int KW>
__global__ __launch_bounds__(KW * kWave, 1) void mha_bwd_fused_short_kernel(MhaBwdParams p) {
  constexpr int D = 64, HD = 32, NT = 2, LS = D + 4, THREADS = KW * kWave, QT = KW, TS = kTile + 4;
  static_assert(KW == 8, "the closing sum of the partial dK / dV gives each of eight waves eight registers");
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *s_q = s_dyn;                     // [QT * 32][68]; tile w becomes wave w's dS^T [32][36] once dK has read it
  float *s_do = s_q + QT * kTile * LS;    // [QT * 32][68]
  float *s_lse = s_do + QT * kTile * LS;  // [QT * 32]
  float *s_delta = s_lse + QT * kTile;    // [QT * 32]
  float *s_k = s_delta + QT * kTile;      // [32][68]: the key tile -- B operand of S and of dQ

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int k0 = th.tile * kTile, mykey = k0 + l31;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

#ifdef CODA_ATTN_PROF
  long long prof_t0 = 0;
#endif
  CODA_PROF_STAMP(0);
  float vf[HD];  // B operand of dP: V[mykey][half*HD + c]  (K's rows come from s_k)
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 b4 = *reinterpret_cast<const float4 *>(vbase + static_cast<size_t>(mykey) * vstride + half * HD + c);
    vf[c] = b4.x; vf[c + 1] = b4.y; vf[c + 2] = b4.z; vf[c + 3] = b4.w;
  }
  static_assert(THREADS == kTile * (D / 4), "one float4 of the key tile per thread");
  const float4 k4 = *reinterpret_cast<const float4 *>(kbase + static_cast<size_t>(k0 + tid / (D / 4)) * kstride + (tid % (D / 4)) * 4);
  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  CODA_PROF_STAMP(1);
  const int nqt = p.l / kTile;
  float *part = p.ds + (static_cast<size_t>(bh) * gridDim.x + th.tile) * nqt * (kTile * D) + lane;
  const float sscale = p.scale * kLog2e;
  const uint32_t qoff = static_cast<uint32_t>(tid / (D / 4)) * static_cast<uint32_t>(qstride) + (tid % (D / 4)) * 4;
  const uint32_t roff = static_cast<uint32_t>(tid / (D / 4)) * static_cast<uint32_t>(rstride) + (tid % (D / 4)) * 4;
  for (int qbase0 = 0; qbase0 < p.l; qbase0 += kTile * QT) {
    if (qbase0) lds_only_barrier();  // every wave is done with the previous step's tiles (its stores stay in flight)
    // stage QT query tiles of Q and dO; delta[row] = sum(dO * O) from the 16 threads that copy the row
    // two batches of 12 loads per thread (one batch of 24 spills); the first one is in flight together with the K / V
    // loads above: the key tile goes to LDS once that batch has been issued
    constexpr int NJ = QT * kTile * (D / 4) / THREADS / 2;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      float4 q4[NJ], g4[NJ], o4[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {  // rows 32 (jb NJ + j) + tid / 16 of the step: a uniform base + the thread's own offset
        const int row0 = qbase0 + (jb * NJ + j) * (THREADS / (D / 4));
        q4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        g4[j] = q4[j];
        o4[j] = q4[j];
        if (row0 < p.l) {  // whole tiles: uniform
          q4[j] = *reinterpret_cast<const float4 *>(qbase + static_cast<size_t>(row0) * qstride + qoff);
          g4[j] = *reinterpret_cast<const float4 *>(p.dout + head_off + static_cast<size_t>(row0) * rstride + roff);
          o4[j] = *reinterpret_cast<const float4 *>(p.out + head_off + static_cast<size_t>(row0) * rstride + roff);
        }
      }
      if (jb == 0 && qbase0 == 0) *reinterpret_cast<float4 *>(s_k + (tid / (D / 4)) * LS + (tid % (D / 4)) * 4) = k4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = tid + (jb * NJ + j) * THREADS, row = i / (D / 4), c4 = i % (D / 4);
        *reinterpret_cast<float4 *>(s_q + row * LS + c4 * 4) = q4[j];
        *reinterpret_cast<float4 *>(s_do + row * LS + c4 * 4) = g4[j];
        float part_d = g4[j].x * o4[j].x + g4[j].y * o4[j].y + g4[j].z * o4[j].z + g4[j].w * o4[j].w;
        part_d += __shfl_xor(part_d, 1, kWave);
        part_d += __shfl_xor(part_d, 2, kWave);
        part_d += __shfl_xor(part_d, 4, kWave);
        part_d += __shfl_xor(part_d, 8, kWave);
        if (c4 == 0) s_delta[row] = part_d;
      }
      __builtin_amdgcn_sched_barrier(0);  // the second batch's loads stay behind the first batch's stores (registers)
    }
    CODA_PROF_STAMP(2);
    if (tid < kTile * QT) {
      const int qq = qbase0 + tid;
      s_lse[tid] = qq < p.l ? p.lse[static_cast<size_t>(bh) * p.l + qq] * kLog2e : 0.f;
    }
    lds_only_barrier();
    CODA_PROF_STAMP(3);
    const int q0 = qbase0 + w * kTile;
    if (q0 < p.l) {
      float *tq = s_q + w * kTile * LS;
      const float *tdo = s_do + w * kTile * LS, *t_lse = s_lse + w * kTile, *t_delta = s_delta + w * kTile;
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 qa = *reinterpret_cast<const float4 *>(tq + l31 * LS + half * HD + c);
        const float4 ga = *reinterpret_cast<const float4 *>(tdo + l31 * LS + half * HD + c);
        const float4 ka = *reinterpret_cast<const float4 *>(s_k + l31 * LS + half * HD + c);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, ka.x, sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, vf[c], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, ka.y, sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, vf[c + 1], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, ka.z, sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, vf[c + 2], pacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, ka.w, sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, vf[c + 3], pacc, 0, 0, 0);
      }
      CODA_PROF_STAMP(4);
      float pd[16], ds[16];
      {
        const int par = l31 & 1;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float keep0 = 1.f, keep1 = 1.f;
          if (use_drop) {
            const uint32_t mine = drop_hash(dconst, q0 + crow(r, half) + par, p.s, mykey);
            const uint32_t other = __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
            keep0 = drop_keep(par ? other : mine, mykey, p.thresh16) ? p.inv_keep : 0.f;
            keep1 = drop_keep(par ? mine : other, mykey, p.thresh16) ? p.inv_keep : 0.f;
          }
          const int qi = crow(r, half);
          const f32x2 lse2 = {t_lse[qi], t_lse[qi + 1]}, del2 = {t_delta[qi], t_delta[qi + 1]};
          const f32x2 keep2 = {keep0, keep1}, s2 = {sacc[r], sacc[r + 1]}, dp2 = {pacc[r], pacc[r + 1]};
          const f32x2 arg = __builtin_elementwise_fma(s2, f32x2{sscale, sscale}, -lse2);
          const f32x2 prob = {fast_exp2(arg[0]), fast_exp2(arg[1])};
          const f32x2 pdv = prob * keep2;
          const f32x2 dsv = prob * __builtin_elementwise_fma(dp2, keep2, -del2);
          pd[r] = pdv[0]; pd[r + 1] = pdv[1];
          ds[r] = dsv[0]; ds[r + 1] = dsv[1];
        }
      }
      CODA_PROF_STAMP(5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = crow(r, half);
        const float2 g2 = *reinterpret_cast<const float2 *>(tdo + qi * LS + NT * l31);
        const float2 q2 = *reinterpret_cast<const float2 *>(tq + qi * LS + NT * l31);
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.x, dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[r], g2.y, dv[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.x, dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], q2.y, dk[1], 0, 0, 0);
      }
      CODA_PROF_STAMP(6);
      // dS^T into the wave's own Q tile (every read of it has been issued; LDS operations of a wave complete in order)
#pragma unroll
      for (int r = 0; r < 16; ++r) tq[crow(r, half) * TS + l31] = ds[r];
      // dQ[q][c] = sum_key dS[q][key] K[key][c]: A = dS^T rows (lane = query), k-slot (kk, half) <-> key 16 half + kk
      f32x16 dq[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) {
        const float4 a4 = *reinterpret_cast<const float4 *>(tq + l31 * TS + 16 * half + kk);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 b2 = *reinterpret_cast<const float2 *>(s_k + (16 * half + kk + i) * LS + NT * l31);
          dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b2.x, dq[0], 0, 0, 0);
          dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b2.y, dq[1], 0, 0, 0);
        }
      }
      CODA_PROF_STAMP(7);
      float *pt = part + static_cast<size_t>(q0 / kTile) * (kTile * D);  // [t][r][lane]
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[(t * 16 + r) * kWave] = dq[t][r];
    }
  }

  CODA_PROF_STAMP(8);
  // the waves' partial dK / dV: [wave][t * 16 + r | 32 + t * 16 + r][64 lanes] through LDS; wave w then owns dK (w < 4) or
  // dV (w >= 4), registers r = 4 (w & 3) .. + 3 of both column tiles, and adds the eight partials in wave order
  lds_only_barrier();
  {
    float *slot = s_dyn + static_cast<size_t>(w) * (2 * NT * 16) * kWave + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        slot[(t * 16 + r) * kWave] = dk[t][r];
        slot[((NT + t) * 16 + r) * kWave] = dv[t][r];
      }
  }
  lds_only_barrier();
  CODA_PROF_STAMP(9);
  {
    const int which = w >> 2, r0 = 4 * (w & 3);
    const float mul = which ? 1.f : p.scale;
    float *obase = which ? p.dv : p.dk;
    const int ldo = which ? p.lddv : p.lddk;
#pragma unroll
    for (int r = r0; r < r0 + 4; ++r) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int ww = 0; ww < KW; ++ww) {
        const float *sl = s_dyn + static_cast<size_t>(ww) * (2 * NT * 16) * kWave + (which * NT * 16 + r) * kWave + lane;
        a0 += sl[0];
        a1 += sl[16 * kWave];
      }
      const int key = k0 + crow(r, half);
      float *orow = obase + (static_cast<size_t>(key) * p.b + bi) * ldo + hi * D + NT * l31;
      *reinterpret_cast<float2 *>(orow) = make_float2(a0 * mul, a1 * mul);
    }
  }
  CODA_PROF_STAMP(10);
}

// dQ = scale * (sum over the key blocks, in order) of the one-kernel backward's partial tiles.  One workgroup per
// (query tile, head).  LAYOUT 0 (mha_bwd_fused_kernel): element ((w * 2 + qh) * 4 + r) * 64 + lane = query 16 qh +
// 4 (lane >> 4) + r, component 16 w + (lane & 15); a thread owns four consecutive lanes = four consecutive components.
// LAYOUT 1 (mha_bwd_fused_short_kernel): element (t * 16 + r) * 64 + lane = query crow(r, lane >> 5), component
// 2 (lane & 31) + t; a thread owns both t of one (r, lane).
template <int LAYOUT>
__global__ __launch_bounds__(256) void mha_dq_reduce_kernel(MhaBwdParams p, int nkb) {
  constexpr int D = 64, TILE = kTile * D;
  const int qt = blockIdx.x, bh = blockIdx.y, bi = bh / p.h, hi = bh % p.h, nqt = gridDim.x;
  const float *part = p.ds + (static_cast<size_t>(bh) * nkb * nqt + qt) * TILE;
  const size_t kbs = static_cast<size_t>(nqt) * TILE;
  if constexpr (LAYOUT == 0) {
#pragma unroll
    for (int i = 0; i < TILE / 4 / 256; ++i) {
      const int e = 4 * (threadIdx.x + 256 * i);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int kb = 0; kb < nkb; kb += 8) {  // eight loads in flight, summed in key-block order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = kb + u < nkb ? *reinterpret_cast<const float4 *>(part + (kb + u) * kbs + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      const int ln = e & 63, r = (e >> 6) & 3, qh = (e >> 8) & 1, w = e >> 9;
      const int qq = qt * kTile + 16 * qh + 4 * (ln >> 4) + r, c = 16 * w + (ln & 15);
      float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + c;
      *reinterpret_cast<float4 *>(row) = make_float4(acc.x * p.scale, acc.y * p.scale, acc.z * p.scale, acc.w * p.scale);
    }
  } else {
#pragma unroll
    for (int i = 0; i < TILE / 2 / 256; ++i) {
      const int e = threadIdx.x + 256 * i;  // (r, lane)
      float a0 = 0.f, a1 = 0.f;
      for (int kb = 0; kb < nkb; kb += 8) {
        float v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v0[u] = kb + u < nkb ? part[(kb + u) * kbs + e] : 0.f;
          v1[u] = kb + u < nkb ? part[(kb + u) * kbs + 16 * 64 + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += v0[u]; a1 += v1[u]; }
      }
      const int ln = e & 63, r = e >> 6;
      const int qq = qt * kTile + crow(r, ln >> 5);
      float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + 2 * (ln & 31);
      *reinterpret_cast<float2 *>(row) = make_float2(a0 * p.scale, a1 * p.scale);
    }
  }
}

// delta = rowsum(dO * O) of this lane's query inside the dQ kernel (fuse_delta): the lane holds its half of the dO row
// already; the other half comes from lane ^ 32.  Written once per query for the dK/dV kernel that runs behind this one
// -- a launch (6.5 us, 19 per step) less than the stand-alone mha_delta_kernel.
template <int HD>
__device__ __forceinline__ float row_delta(const MhaBwdParams &p, const float (&gf)[HD], int myq, int bh, size_t rstride,
                                           size_t head_off, int half, bool writer) {
  float acc = 0.f;
  if (myq < p.l) {
    const float *orow = p.out + static_cast<size_t>(myq) * rstride + head_off + half * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 o = *reinterpret_cast<const float4 *>(orow + c);
      acc += o.x * gf[c] + o.y * gf[c + 1] + o.z * gf[c + 2] + o.w * gf[c + 3];
    }
  }
  acc += __shfl_xor(acc, 32, kWave);
  if (writer && half == 0 && myq < p.l) p.delta[static_cast<size_t>(bh) * p.l + myq] = acc;
  return acc;
}

// dQ: a wave owns 32 queries, loops over key tiles.  S^T / dP^T are evaluated transposed as in
// the forward (lane = query, registers = keys), which is the A-operand layout of dQ = dS K.
// QT: see mha_fwd_kernel.  KH (SPLIT only): workgroups per query-tile group, each over 1 / KH of the keys -- 256 queries
// give 4 workgroups of two query tiles per head, half the chip; with KH = 2 the chip is full again, every staged K / V
// tile still serves two query tiles, and the two partial dQ meet in float atomics on a zeroed dQ (two addends: the sum
// does not depend on their order).
template <int D, int QW, bool SPLIT, bool GEN, bool DB = false, int QT = 1, int KH = 1>
__global__ __launch_bounds__(QW * QT * kWave, ((DB && !SPLIT && D == 64) ? 2 : 1)) void mha_bwd_dq_kernel(MhaBwdParams p) {
  static_assert(QT == 1 || SPLIT, "several query tiles per workgroup: split-key form only");
  static_assert(KH == 1 || (SPLIT && DB), "key halves: double-buffered split-key form only");
  constexpr int HD = D / 2, NT = D / 32, LS = D + 4, THREADS = QW * QT * kWave;
  constexpr int TILES = (DB && !SPLIT) ? QW / 2 : QW;  // DB: two stages, double buffered (see mha_fwd_kernel)
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *s_k = s_dyn;
  float *s_v = s_dyn + TILES * kTile * LS;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int qt = SPLIT ? w / QW : 0, ks = SPLIT ? w % QW : w;
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int kh = KH > 1 ? th.tile % KH : 0, qgroup = KH > 1 ? th.tile / KH : th.tile;
  const int q0 = SPLIT ? (qgroup * QT + qt) * kTile : (th.tile * QW + w) * kTile;
  const int my_tile = SPLIT ? ks : 0;
  const int s_begin = KH > 1 ? kh * (p.s / KH) : 0, s_end = KH > 1 ? s_begin + p.s / KH : p.s;  // (host: S % (KH * stage) == 0)
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

  float qf[HD], gf[HD];
  const float qscale = p.scale * kLog2e;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = a;
    if (myq < p.l) {
      a = *reinterpret_cast<const float4 *>(qbase + static_cast<size_t>(myq) * qstride + half * HD + c);
      g = *reinterpret_cast<const float4 *>(p.dout + static_cast<size_t>(myq) * rstride + head_off + half * HD + c);
    }
    // scale * log2(e) folded into Q (S in log2 units, as in the forward)
    qf[c] = a.x * qscale; qf[c + 1] = a.y * qscale; qf[c + 2] = a.z * qscale; qf[c + 3] = a.w * qscale;
    gf[c] = g.x; gf[c + 1] = g.y; gf[c + 2] = g.z; gf[c + 3] = g.w;
  }
  float lse = 0.f, delta = 0.f;
  if (myq < p.l) {
    lse = p.lse[static_cast<size_t>(bh) * p.l + myq] * kLog2e;  // log2 units
    if (!p.fuse_delta) delta = p.delta[static_cast<size_t>(bh) * p.l + myq];
  }
  if (p.fuse_delta) delta = row_delta<HD>(p, gf, myq, bh, rstride, head_off, half, !SPLIT || (ks == 0 && kh == 0));
  // rows past the end or without any admissible key: exp2(x - inf) = 0 without a per-element test
  const float lse_eff = (myq < p.l && lse != -INFINITY) ? lse : INFINITY;
  f32x16 dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  constexpr int kStageFloats = 2 * TILES * kTile * LS;  // one K + V stage
  constexpr int NLD = DB ? TILES * kTile * D / 4 / THREADS : 1;
  constexpr int FROWS = DB ? TILES * kTile : THREADS * 4 / D;
  float4 rk[NLD], rv[NLD];
  if (DB) {  // prologue: stage 0 straight into buffer 0
    fetch_tile<D, THREADS, FROWS>(rk, kbase, kstride, s_begin, p.s, tid);
    fetch_tile<D, THREADS, FROWS>(rv, vbase, vstride, s_begin, p.s, tid);
    store_tile<D, THREADS, FROWS>(s_k, rk, tid);
    store_tile<D, THREADS, FROWS>(s_v, rv, tid);
    __syncthreads();
  }
  int stage = 0;
  for (int sbase = s_begin; sbase < s_end; sbase += kTile * TILES, ++stage) {
    const bool more = DB && sbase + kTile * TILES < s_end;
    if (!DB) {
      __syncthreads();
      load_tile<D, THREADS, kTile * TILES>(s_k, kbase, kstride, sbase, p.s, tid);
      load_tile<D, THREADS, kTile * TILES>(s_v, vbase, vstride, sbase, p.s, tid);
      __syncthreads();
    } else {
      s_k = s_dyn + (stage & 1) * kStageFloats;
      s_v = s_k + TILES * kTile * LS;
      if (more) {  // next stage: loads in flight during this stage's MFMAs
        fetch_tile<D, THREADS, FROWS>(rk, kbase, kstride, sbase + kTile * TILES, p.s, tid);
        fetch_tile<D, THREADS, FROWS>(rv, vbase, vstride, sbase + kTile * TILES, p.s, tid);
      }
    }
    for (int tile = SPLIT ? my_tile : 0; tile < (SPLIT ? my_tile + 1 : TILES); ++tile) {
    const int s0 = sbase + tile * kTile;
    if (!wave_active || s0 >= p.s) break;
    const float *tk = s_k + tile * kTile * LS, *tv = s_v + tile * kTile * LS;

    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 ka = *reinterpret_cast<const float4 *>(tk + l31 * LS + half * HD + c);
      const float4 va = *reinterpret_cast<const float4 *>(tv + l31 * LS + half * HD + c);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qf[c], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, gf[c], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qf[c + 1], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, gf[c + 1], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qf[c + 2], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, gf[c + 2], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qf[c + 3], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, gf[c + 3], pacc, 0, 0, 0);
    }
    float ds[16];
    constexpr bool plain = !GEN;  // no mask, L and S multiples of the tile; lse_eff = +inf kills empty rows
    if (plain) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float keep0 = 1.f, keep1 = 1.f;
        if (use_drop) {
          const uint32_t hsh = drop_hash(dconst, myq, p.s, s0 + crow(r, half));
          keep0 = drop_keep_lo(hsh, p.thresh16) ? p.inv_keep : 0.f;
          keep1 = drop_keep_hi(hsh, p.thresh16) ? p.inv_keep : 0.f;
        }
        const f32x2 arg = f32x2{sacc[r], sacc[r + 1]} - f32x2{lse_eff, lse_eff};  // packed fp32 (v_pk_*)
        const f32x2 prob = {fast_exp2(arg[0]), fast_exp2(arg[1])};
        const f32x2 dsv = prob * __builtin_elementwise_fma(f32x2{pacc[r], pacc[r + 1]}, f32x2{keep0, keep1},
                                                            f32x2{-delta, -delta});  // * scale: once, on the dQ rows
        ds[r] = dsv[0]; ds[r + 1] = dsv[1];
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
        ds[r] = prob * __fmaf_rn(pacc[r], keep, -delta);
      }
    }
    // dQ[q][d] += sum_key dS[q][key] K[key][d]   (A = dS: lane = query, k = key)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *krow = tk + crow(r, half) * LS + NT * l31;
      if (NT == 2) {
        const float2 k2 = *reinterpret_cast<const float2 *>(krow);
        dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], k2.x, dq[0], 0, 0, 0);
        dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], k2.y, dq[1], 0, 0, 0);
      } else {
        const float4 k4 = *reinterpret_cast<const float4 *>(krow);
        dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], k4.x, dq[0], 0, 0, 0);
        dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], k4.y, dq[1], 0, 0, 0);
        dq[2 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], k4.z, dq[2 % NT], 0, 0, 0);
        dq[3 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], k4.w, dq[3 % NT], 0, 0, 0);
      }
    }
    }  // tile
    if (DB) {
      if (more) {
        float *nk = s_dyn + ((stage + 1) & 1) * kStageFloats;
        store_tile<D, THREADS, FROWS>(nk, rk, tid);
        store_tile<D, THREADS, FROWS>(nk + TILES * kTile * LS, rv, tid);
      }
      __syncthreads();
    }
  }
  if (SPLIT && QW > 1) {  // sum the per-wave partial dQ through LDS: [wave-1][NT*16][64 lanes]
    __syncthreads();
    float *sq = s_dyn + static_cast<size_t>(qt) * (QW - 1) * (NT * 16) * kWave;  // this query tile's slots
    if (ks > 0) {
      float *slot = sq + static_cast<size_t>(ks - 1) * (NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * kWave + lane] = dq[t][r];
    }
    __syncthreads();
    if (ks > 0) return;
    for (int ww = 1; ww < QW; ++ww) {
      const float *sl = sq + static_cast<size_t>(ww - 1) * (NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] += sl[(t * 16 + r) * kWave + lane];
    }
  }
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + crow(r, half);
      if (qq < p.l) {
        float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + NT * l31;
        if (KH > 1) {
#pragma unroll
          for (int t = 0; t < NT; ++t) unsafeAtomicAdd(row + t, dq[t][r] * p.scale);
        } else if (NT == 2) {
          *reinterpret_cast<float2 *>(row) = make_float2(dq[0][r] * p.scale, dq[1][r] * p.scale);
        } else {
          *reinterpret_cast<float4 *>(row) = make_float4(dq[0][r] * p.scale, dq[1][r] * p.scale, dq[2 % NT][r] * p.scale,
                                                          dq[3 % NT][r] * p.scale);
        }
      }
    }
  }
}

// Split-key dQ WITHOUT LDS staging (see mha_fwd_direct_kernel): a wave's K / V tiles are its own, so their
// fragments are loaded from L2 straight into the three register layouts the MFMAs consume -- K rows (A operand of
// S^T = K Q^T), V rows (A operand of dP^T = V dO^T), K columns (B operand of dQ = dS K) -- one tile ahead of the
// MFMAs; LDS only for the final sum of the QW partial dQ.
template <int D, int QW, bool GEN>
__global__ __launch_bounds__(QW * kWave) void mha_bwd_dq_direct_kernel(MhaBwdParams p) {
  constexpr int HD = D / 2, NT = D / 32;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = th.tile * kTile;
  const int myq = q0 + l31;
  const size_t rstride = static_cast<size_t>(p.b) * p.h * D;
  const size_t head_off = (static_cast<size_t>(bi) * p.h + hi) * D;
  const bool use_drop = p.thresh16 != 0u;
  const uint32_t dconst = use_drop ? drop_const(effective_seed(p.seed, p.seed_dev), static_cast<uint32_t>(bh)) : 0u;
  const size_t qstride = static_cast<size_t>(p.b) * p.ldq, kstride = static_cast<size_t>(p.b) * p.ldk,
               vstride = static_cast<size_t>(p.b) * p.ldv;
  const float *qbase = p.q + static_cast<size_t>(bi) * p.ldq + hi * D;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *vbase = p.v + static_cast<size_t>(bi) * p.ldv + hi * D;

  float qf[HD], gf[HD];
  const float qscale = p.scale * kLog2e;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = a;
    if (myq < p.l) {
      a = *reinterpret_cast<const float4 *>(qbase + static_cast<size_t>(myq) * qstride + half * HD + c);
      g = *reinterpret_cast<const float4 *>(p.dout + static_cast<size_t>(myq) * rstride + head_off + half * HD + c);
    }
    qf[c] = a.x * qscale; qf[c + 1] = a.y * qscale; qf[c + 2] = a.z * qscale; qf[c + 3] = a.w * qscale;
    gf[c] = g.x; gf[c + 1] = g.y; gf[c + 2] = g.z; gf[c + 3] = g.w;
  }
  float lse = 0.f, delta = 0.f;
  if (myq < p.l) {
    lse = p.lse[static_cast<size_t>(bh) * p.l + myq] * kLog2e;  // log2 units
    if (!p.fuse_delta) delta = p.delta[static_cast<size_t>(bh) * p.l + myq];
  }
  if (p.fuse_delta) delta = row_delta<HD>(p, gf, myq, bh, rstride, head_off, half, w == 0);
  const float lse_eff = (myq < p.l && lse != -INFINITY) ? lse : INFINITY;
  f32x16 dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  struct Frag {
    float4 k[HD / 4], v[HD / 4];  // rows l31 of the tile, half `half` of the head dimension
    float kc[16][NT];             // K[crow(r, half)][NT * l31 ..]
  };
  auto load_kv = [&](Frag &f, int s0) {
    const int row = s0 + l31;
    const bool in = !GEN || row < p.s;
    const float *kp = kbase + static_cast<size_t>(in ? row : 0) * kstride + half * HD;
    const float *vp = vbase + static_cast<size_t>(in ? row : 0) * vstride + half * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const float4 a = *reinterpret_cast<const float4 *>(kp + 4 * c), b = *reinterpret_cast<const float4 *>(vp + 4 * c);
      f.k[c] = in ? a : make_float4(0.f, 0.f, 0.f, 0.f);
      f.v[c] = in ? b : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load_kc = [&](Frag &f, int s0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = s0 + crow(r, half);
      const bool kin = !GEN || krow < p.s;
      const float *cp = kbase + static_cast<size_t>(kin ? krow : 0) * kstride + NT * l31;
      if (NT == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(cp);
        f.kc[r][0] = kin ? t.x : 0.f; f.kc[r][1] = kin ? t.y : 0.f;
      } else {
        const float4 t = *reinterpret_cast<const float4 *>(cp);
        f.kc[r][0] = kin ? t.x : 0.f; f.kc[r][1] = kin ? t.y : 0.f; f.kc[r][2 % NT] = kin ? t.z : 0.f; f.kc[r][3 % NT] = kin ? t.w : 0.f;
      }
    }
  };
  float ds[16];
  auto scores = [&](const Frag &f, int s0) {
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 ka = f.k[c / 4], va = f.v[c / 4];
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qf[c], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, gf[c], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qf[c + 1], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, gf[c + 1], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qf[c + 2], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, gf[c + 2], pacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qf[c + 3], sacc, 0, 0, 0);
      pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, gf[c + 3], pacc, 0, 0, 0);
    }
    if (!GEN) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float keep0 = 1.f, keep1 = 1.f;
        if (use_drop) {
          const uint32_t hsh = drop_hash(dconst, myq, p.s, s0 + crow(r, half));
          keep0 = drop_keep_lo(hsh, p.thresh16) ? p.inv_keep : 0.f;
          keep1 = drop_keep_hi(hsh, p.thresh16) ? p.inv_keep : 0.f;
        }
        const f32x2 arg = f32x2{sacc[r], sacc[r + 1]} - f32x2{lse_eff, lse_eff};  // packed fp32 (v_pk_*)
        const f32x2 prob = {fast_exp2(arg[0]), fast_exp2(arg[1])};
        const f32x2 dsv = prob * __builtin_elementwise_fma(f32x2{pacc[r], pacc[r + 1]}, f32x2{keep0, keep1},
                                                            f32x2{-delta, -delta});  // * scale: once, on the dQ rows
        ds[r] = dsv[0]; ds[r + 1] = dsv[1];
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
        ds[r] = prob * __fmaf_rn(pacc[r], keep, -delta);
      }
    }
  };
  auto accumulate = [&](const Frag &f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int t = 0; t < NT; ++t) dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[r], f.kc[r][t], dq[t], 0, 0, 0);
    }
  };

  const int ntile = (p.s + kTile - 1) / kTile;
  const bool wave_active = q0 < p.l;
  if (wave_active && D <= 64) {  // two register sets: the whole next tile in flight during this tile's MFMAs
    // (unconditional prefetch, the last one re-reads the final tile: see mha_fwd_direct_kernel)
    Frag fa, fb;
    int t = w;
    const int last = ntile - 1;
    if (t < ntile) {
      load_kv(fa, t * kTile);
      load_kc(fa, t * kTile);
      while (true) {
        load_kv(fb, min(t + QW, last) * kTile);
        load_kc(fb, min(t + QW, last) * kTile);
        scores(fa, t * kTile);
        accumulate(fa);
        t += QW;
        if (t >= ntile) break;
        load_kv(fa, min(t + QW, last) * kTile);
        load_kc(fa, min(t + QW, last) * kTile);
        scores(fb, t * kTile);
        accumulate(fb);
        t += QW;
        if (t >= ntile) break;
      }
    }
  } else if (wave_active) {
    // head width 128: 192 registers per fragment set -- one set, phased: the K columns of the CURRENT tile load
    // under its S / dP MFMAs (128 of them), the K / V rows of the NEXT tile under its dQ MFMAs (64) + soft-max
    Frag f;
    int t = w;
    const int last = ntile - 1;
    if (t < ntile) load_kv(f, t * kTile);
    while (t < ntile) {
      load_kc(f, t * kTile);
      scores(f, t * kTile);
      load_kv(f, min(t + QW, last) * kTile);  // (unconditional: exact waits in front of the dQ MFMAs)
      accumulate(f);
      t += QW;
    }
  }
  if (QW > 1) {  // sum the per-wave partial dQ through LDS: [wave-1][NT*16][64 lanes]
    if (w > 0) {
      float *slot = s_dyn + static_cast<size_t>(w - 1) * (NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * kWave + lane] = dq[t][r];
    }
    __syncthreads();
    if (w > 0) return;
    for (int ww = 1; ww < QW; ++ww) {
      const float *sl = s_dyn + static_cast<size_t>(ww - 1) * (NT * 16) * kWave;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] += sl[(t * 16 + r) * kWave + lane];
    }
  }
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + crow(r, half);
      if (qq < p.l) {
        float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + NT * l31;
        if (NT == 2) {
          *reinterpret_cast<float2 *>(row) = make_float2(dq[0][r] * p.scale, dq[1][r] * p.scale);
        } else {
          *reinterpret_cast<float4 *>(row) = make_float4(dq[0][r] * p.scale, dq[1][r] * p.scale, dq[2 % NT][r] * p.scale,
                                                          dq[3 % NT][r] * p.scale);
        }
      }
    }
  }
}

uint32_t drop_threshold(float p) {  // 16-bit: keep iff hash16 >= threshold; 0 = dropout off
  if (!(p > 0.f)) return 0u;
  double t = static_cast<double>(p) * 65536.0 + 0.5;
  if (t < 1.0) t = 1.0;
  if (t > 65535.0) t = 65535.0;
  return static_cast<uint32_t>(t);
}

// Raise the dynamic-LDS limit of a kernel once per process (idempotent; kept out of the
// steady-state launch path so that launches can be captured into hipGraphs).
template <typename K>
int set_lds(K kern, size_t bytes) {
  return raise_dynamic_lds(kern, bytes);  // per (kernel entry point, device), common.hip.h
}

// double-buffered K/V staging of the long-sequence kernels (CODA_ATTN_DB=0: single buffer, A/B)
// split-key kernels: 4 waves with two full stages in LDS (default) or 8 waves, single stage
// (CODA_ATTN_SPLIT_DB=0, A/B): measured 53 -> 40 us forward, 65 -> 50 us dQ on the decoder shapes
bool split_double_buffered() {
  static const bool on = [] { const char *e = getenv("CODA_ATTN_SPLIT_DB"); return !e || atoi(e) != 0; }();
  return on;
}
// split-key kernels without LDS staging (CODA_ATTN_DIRECT = 0 off | 4 | 8 waves per workgroup; default 4)
int split_direct() {
  static const int v = [] { const char *e = getenv("CODA_ATTN_DIRECT"); return e ? atoi(e) : 4; }();
  return v;
}
// ---- optional per-kernel HIP-event timing (coda_mha_timing_*) -----------------------------------
struct TimingRecord {
  int kind, l, s;
  hipEvent_t e0, e1;
};
int g_timing_min_len = -1;  // < 0: off
unsigned g_timing_kinds = ~0u;  // bit k: kernels of kind k are recorded (coda_mha_timing_enable_kinds)
std::vector<TimingRecord> g_timing;
constexpr size_t kTimingCap = 16384;

void timing_clear() {
  for (auto &r : g_timing) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_timing.clear();
}

// brackets the launches issued during its lifetime
// dQ = scale * dS K as a plain GEMM per (scene, head): (L x S) . (S x D), dS from the workspace the dK/dV kernel filled.
// A workgroup = 4 waves x 32 queries; the contraction runs over 64-key chunks.  A fragments come straight from memory:
// the MFMA's k-slot (kk, half) is mapped to key 8 (kk / 4) + 4 half + kk % 4, so a lane's four consecutive k-steps are
// one 16-byte load of its own query row and nothing is fetched twice; the K chunk (64 keys x D) goes through LDS,
// double buffered, and is read as B[k-slot][column] = K[key][32 cb + lane & 31] (32 consecutive floats per half-wave).
template <int D, int KC>
__global__ __launch_bounds__(256, 2) void mha_bwd_dq_gemm_kernel(MhaBwdParams p) {
  constexpr int NT = D / 32, NLD = KC * D / 4 / 256;  // float4 per thread and K chunk
  extern __shared__ __attribute__((aligned(16))) float s_kdyn[];
  float (*s_k)[KC * D] = reinterpret_cast<float (*)[KC * D]>(s_kdyn);
  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = (th.tile * 4 + w) * kTile;
  const bool wave_active = q0 < p.l;
  const size_t kstride = static_cast<size_t>(p.b) * p.ldk;
  const float *kbase = p.k + static_cast<size_t>(bi) * p.ldk + hi * D;
  const float *arow = p.ds + (static_cast<size_t>(bh) * p.l + (wave_active ? q0 + l31 : 0)) * p.s + 4 * half;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // K (512 KB per head, shared by the head's 16 workgroups: L2 hits) is fetched one chunk ahead; dS -- the 537 MB stream
  // from HBM, every byte used once -- TWO chunks ahead (one chunk of MFMAs, 1.7 us, does not cover an HBM round trip).
  float4 rk[NLD], ra0[KC / 8], ra1[KC / 8];
  auto fetch_k = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i, key = e / (D / 4), c4 = e % (D / 4);
      rk[i] = c0 + key < p.s ? *reinterpret_cast<const float4 *>(kbase + static_cast<size_t>(c0 + key) * kstride + 4 * c4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto fetch_a = [&](float4 (&ra)[KC / 8], int c0) {
#pragma unroll
    for (int j = 0; j < KC / 8; ++j)
      // plain (cached) loads: a wave instruction takes 32 bytes of each of its 32 rows, so every 128-byte line is
      // touched by four consecutive instructions -- with streaming (non-temporal) loads the kernel took 0.244 ms, with
      // the lines kept in L1 0.197
      ra[j] = (wave_active && c0 + 8 * j + 4 * half < p.s) ? *reinterpret_cast<const float4 *>(arow + c0 + 8 * j)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  fetch_k(0);
  fetch_a(ra0, 0);
  fetch_a(ra1, KC);
  int stage = 0;
  for (int c0 = 0; c0 < p.s; c0 += KC, stage ^= 1) {
    float *sk = s_k[stage];
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<float4 *>(sk + 4 * (tid + 256 * i)) = rk[i];
    float4 a[KC / 8];
#pragma unroll
    for (int j = 0; j < KC / 8; ++j) { a[j] = ra0[j]; ra0[j] = ra1[j]; }
    lds_only_barrier();  // LDS only: the loads in flight stay in flight (two K stages: the slower waves may still read
                         // this stage's predecessor)
    if (c0 + KC < p.s) fetch_k(c0 + KC);
    fetch_a(ra1, c0 + 2 * KC);  // (past the end: predicated off)
    if (wave_active) {
#pragma unroll
      for (int j = 0; j < KC / 8; ++j) {
        const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float *brow = sk + (8 * j + 4 * half + i) * D + l31;
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], brow[32 * t], acc[t], 0, 0, 0);
        }
      }
    }
  }
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + crow(r, half);
      if (qq < p.l) {
        float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) row[32 * t] = acc[t][r] * p.scale;
      }
    }
  }
}


// ---- dQ = scale * dS K on the bf16 matrix cores at fp32 accuracy (round 6) ------------------------------------------
// The dS K GEMM above reads its 537 MB once and runs at 60 % of the fp32 MFMA peak: neither bound.  With both operands
// in three bf16 pieces the matrix work is 6/16 of that and the kernel becomes the HBM stream it should be.
//  * K^T pieces are made ONCE per call by mha_kt_pieces_kernel: [head][chunk of 64 keys][piece][component][key slot],
//    24 KB per chunk in one run -- the B operand wants 8 consecutive keys of one component per lane.  Inside a 16-key
//    block the slots are ordered (keys 0-3, 8-11 | 4-7, 12-15) so that the two halves of the wave read ADJACENT 16-byte
//    pieces of a dS row with each load instruction.  The pieces of the odd 16-key blocks are stored NEGATED.
//  * dS fragments come straight from memory, two chunks ahead, and are split by the wave that multiplies them (every
//    element of dS is used by one wave only: nothing to share through LDS).
//  * Two accumulator sets: X takes the even 16-key blocks, Y the (negated) odd ones, dQ = scale * (X - Y) -- the bf16
//    matrix core's accumulate truncates toward -inf (gemm_x3.hip): both sets carry the same drift and the difference
//    keeps only its fluctuation.
// A workgroup = 4 waves x 32 queries; K^T chunks double-buffered in LDS (rows 144 B apart), one LDS-only barrier per chunk.
constexpr int kDqChunkBytes = 3 * 64 * 64 * 2;    // one K^T piece chunk in memory
constexpr int kDqLdsBuf = 3 * 64 * kPieceRow;     // ... in LDS
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mha_kt_pieces_kernel(MhaBwdParams p, __bf16 *ktp) {
  const int c = blockIdx.x, bh = blockIdx.y, bi = bh / p.h, hi = bh % p.h;
  const int tid = threadIdx.x, key = tid & 63, kk = key & 15, jb = key >> 4;
  const int slot = 16 * jb + 8 * ((kk >> 2) & 1) + (kk & 3) + 4 * (kk >> 3);
  const float sg = (jb & 1) ? -1.f : 1.f;
  const float *krow = p.k + (static_cast<size_t>(c * 64 + key) * p.b + bi) * p.ldk + hi * 64;
  __bf16 *dst = ktp + (static_cast<size_t>(bh) * gridDim.x + c) * (3 * 64 * 64) + slot;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d4 = (tid >> 6) + 4 * i;
    const f32x4 x = *reinterpret_cast<const f32x4 *>(krow + 4 * d4);
    const Bf3x4 sp = split3(x * sg);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(q * 64 + 4 * d4 + e) * 64] = sp.p[q][e];
  }
}

__global__ __launch_bounds__(256, 2) void mha_bwd_dq_x3_kernel(MhaBwdParams p, const unsigned char *ktp) {
  constexpr int D = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_kt[];
  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int half = lane >> 5, l31 = lane & 31;
  const TileHead th = tile_head(p.xcd_map);
  const int bh = th.bh, bi = bh / p.h, hi = bh % p.h;
  const int q0 = (th.tile * 4 + w) * kTile;
  const bool wave_active = q0 < p.l;
  const int nch = p.s / 64;
  const float *arow = p.ds + (static_cast<size_t>(bh) * p.l + (wave_active ? q0 + l31 : 0)) * p.s + 4 * half;
  const unsigned char *kt = ktp + static_cast<size_t>(bh) * nch * kDqChunkBytes + 16 * tid;

  f32x16 X[2], Y[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { X[t][r] = 0.f; Y[t][r] = 0.f; }

  u32x4 rb[6];
  f32x4 r0[8], r1[8], r2[8];  // dS fragments of three consecutive chunks; the loop is unrolled three times so that the
                              // sets rotate by NAME (a copy of a set would have to wait for its loads)
  auto fetch_b = [&](int c) {
#pragma unroll
    for (int i = 0; i < 6; ++i) rb[i] = *reinterpret_cast<const u32x4 *>(kt + static_cast<size_t>(c) * kDqChunkBytes + 4096 * i);
  };
  // block j: keys 16 j + 4 half + (0..3) and 16 j + 8 + 4 half + (0..3).  Unconditional loads (past the end: the last
  // chunk again, unused; an idle wave reads row 0): with a predicate per load hipcc turns every one into a branch and
  // its wait counts drain ALL loads in flight at the top of each iteration -- the two-chunks-ahead prefetch was one
  auto fetch_a = [&](f32x4 (&ra)[8], int c) {
    const float *src = arow + 64 * (c < nch ? c : nch - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) ra[j] = *reinterpret_cast<const f32x4 *>(src + 8 * j);
  };
  const int boff = l31 * kPieceRow + 16 * half;  // this lane's B fragment: component l31 (+ 32 t), k block 0
  // one chunk: K^T pieces of chunk c (in rb) to LDS, barrier, loads of chunk c + 1 (K^T) and c + 2 (dS, into the set
  // the previous step has consumed), then 48 MFMAs on this chunk's dS fragments
  auto step = [&](int c, f32x4 (&a)[8], f32x4 (&nxt)[8]) {
    unsigned char *sb = s_kt + (c & 1) * kDqLdsBuf;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<u32x4 *>(sb + (e >> 3) * kPieceRow + (e & 7) * 16) = rb[i];
    }
    lds_only_barrier();  // LDS only: the loads in flight stay in flight (two buffers: slower waves may still read the other one)
    fetch_b(c + 1 < nch ? c + 1 : c);
    fetch_a(nxt, c + 2);
    if (wave_active) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Bf3x4 lo = split3(a[2 * j]), hi4 = split3(a[2 * j + 1]);
        bf16x8 pa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) pa[q] = __builtin_shufflevector(lo.p[q], hi4.p[q], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8 fb[3];
#pragma unroll
          for (int q = 0; q < 3; ++q)
            fb[q] = *reinterpret_cast<const bf16x8 *>(sb + (q * 64 + 32 * t) * kPieceRow + boff + 32 * j);
          f32x16 acc = (j & 1) ? Y[t] : X[t];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], fb[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], fb[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[2], fb[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], fb[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], fb[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], fb[0], acc, 0, 0, 0);
          if (j & 1) Y[t] = acc; else X[t] = acc;
        }
      }
    }
  };
  fetch_b(0);
  fetch_a(r0, 0);
  fetch_a(r1, 1);
  int c = 0;
  for (; c + 3 <= nch; c += 3) {  // (unrolled six times -- one drain of the loads per six chunks at the loop header instead of per three -- measured the same)
    step(c, r0, r2);
    step(c + 1, r1, r0);
    step(c + 2, r2, r1);
  }
  if (c < nch) { step(c, r0, r2); ++c; }
  if (c < nch) step(c, r1, r0);
  if (wave_active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + crow(r, half);
      if (qq < p.l) {
        float *row = p.dq + (static_cast<size_t>(qq) * p.b + bi) * p.lddq + hi * D + l31;
#pragma unroll
        for (int t = 0; t < 2; ++t) row[32 * t] = (X[t][r] - Y[t][r]) * p.scale;
      }
    }
  }
}


struct KernelTimer {
  bool armed = false;
  KernelTimer(int kind, int l, int s, hipStream_t) {
    if (g_timing_min_len < 0 || l < g_timing_min_len || s < g_timing_min_len || g_timing.size() >= kTimingCap ||
        !((g_timing_kinds >> kind) & 1u))
      return;
    TimingRecord r{kind, l, s, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    PendingTiming &t = pending_timing();  // the scope's kernel launch (mha_launch) takes the pair as its own events
    t.e0 = r.e0;
    t.e1 = r.e1;
    armed = true;
    g_timing.push_back(r);
  }
  ~KernelTimer() {
    if (!armed) return;
    PendingTiming &t = pending_timing();
    if (t.e0) {  // nothing was launched in the scope (an error return): drop the record
      t.e0 = t.e1 = nullptr;
      if (!g_timing.empty()) {
        (void)hipEventDestroy(g_timing.back().e0);
        (void)hipEventDestroy(g_timing.back().e1);
        g_timing.pop_back();
      }
    }
  }
};

// CODA_ATTN_XCD=0: plain (tile, head) grid (A/B)
bool xcd_mapped() {
  static const bool on = [] { const char *e = getenv("CODA_ATTN_XCD"); return !e || atoi(e) != 0; }();
  return on;
}
bool double_buffered() {
  static const bool on = [] { const char *e = getenv("CODA_ATTN_DB"); return !e || atoi(e) != 0; }();
  return on;
}

template <int D, bool GEN>
int launch_fwd_g(const MhaParams &p, hipStream_t s) {
  constexpr size_t kTileBytes = sizeof(float) * 2 * kTile * (D + 4);  // one K + one V tile
  // split-key variants: 8 waves per block when 8 K/V tile pairs fit the 160 KB LDS (D = 64)
  constexpr int SW = (8 * kTileBytes <= 160 * 1024) ? 8 : 4;
  KernelTimer timer(0, p.l, p.s, s);
  if (p.l >= 1024) {
    dim3 grid(ceil_div(p.l, kTile * 4), p.b * p.h);
    if (double_buffered()) {
      auto kern = mha_fwd_kernel<D, 4, false, GEN, true>;
      int st = set_lds(kern, 4 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, grid, dim3(256), 4 * kTileBytes, s, p);
    } else {
      auto kern = mha_fwd_kernel<D, 4, false, GEN>;
      int st = set_lds(kern, 4 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, grid, dim3(256), 4 * kTileBytes, s, p);
    }
  } else {
    dim3 grid(ceil_div(p.l, kTile), p.b * p.h);
    const int direct = split_direct();
    if (D == 64 && p.s >= 1024 && split_query_tiles(ceil_div(p.l, 2 * kTile) * p.b * p.h) == 1) {  // K / V staged once for two query tiles
      auto kern = mha_fwd_kernel<D, 4, true, GEN, true, 2>;
      int st = set_lds(kern, 8 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, 2 * kTile), p.b * p.h), dim3(8 * kWave), 8 * kTileBytes, s, p);
    } else if (direct) {  // K / V fragments straight from L2 into registers, no staging (mha_fwd_direct_kernel)
      constexpr size_t mlds4 = sizeof(float) * 3 * (D / 32 * 16 + 2) * kWave, mlds8 = sizeof(float) * 7 * (D / 32 * 16 + 2) * kWave;
      if (direct == 8 && D == 64) {
        auto kern = mha_fwd_direct_kernel<D, 8, GEN>;
        int st = set_lds(kern, mlds8);
        if (st != CODA_OK) return st;
        mha_launch(kern, grid, dim3(8 * kWave), mlds8, s, p);
      } else {
        auto kern = mha_fwd_direct_kernel<D, 4, GEN>;
        int st = set_lds(kern, mlds4);
        if (st != CODA_OK) return st;
        mha_launch(kern, grid, dim3(4 * kWave), mlds4, s, p);
      }
    } else if (split_double_buffered() && 8 * kTileBytes <= 160 * 1024) {
      auto kern = mha_fwd_kernel<D, 4, true, GEN, true>;  // 4 waves, 2 x 4 tile pairs
      int st = set_lds(kern, 8 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, grid, dim3(4 * kWave), 8 * kTileBytes, s, p);
    } else {
      auto kern = mha_fwd_kernel<D, SW, true, GEN>;
      int st = set_lds(kern, SW * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, grid, dim3(SW * kWave), SW * kTileBytes, s, p);
    }
  }
  return launch_status();
}

// MFMA operand type of this call: 0 fp32 (default), 1 bf16 (attention_bf16.hip), 2 three bf16 pieces per fp32 operand.
// The *_opt entry points pass it as an argument (common.hip.h: CallOptions); the library default is CODA_ATTN_DTYPE.
int default_mfma_dtype() {
  static const int v = [] {
    const char *e = getenv("CODA_ATTN_DTYPE");
    // "bf16" / "1": bf16 operands; "bf16x3" / "x3" / "2": three bf16 pieces per fp32 operand (fp32-level results)
    if (e && (e[0] == '2' || e[0] == 'x' || (e[0] == 'b' && strstr(e, "x3")))) return 2;
    if (e && (e[0] == 'b' || e[0] == '1')) return 1;
    return 0;
  }();
  return v;
}
int mfma_dtype() {
  const int v = call_options().mfma_dtype;
  return v >= 0 && v <= 2 ? v : default_mfma_dtype();
}

template <int D>
int launch_fwd(const MhaParams &p, hipStream_t s) {
  clear_sticky_error();
  if (mfma_dtype() == 1) {
    KernelTimer timer(0, p.l, p.s, s);
    const int st = mha_fwd_bf16(p, D, s);
    return st != CODA_OK ? st : launch_status();
  }
  if (mfma_dtype() == 2 && mha_x3_takes_fwd(p, D)) {
    KernelTimer timer(0, p.l, p.s, s);
    const int st = mha_fwd_x3(p, D, s);
    return st != CODA_OK ? st : launch_status();
  }
  const bool gen = p.mask != nullptr || (p.l % kTile) != 0 || (p.s % kTile) != 0;
  return gen ? launch_fwd_g<D, true>(p, s) : launch_fwd_g<D, false>(p, s);
}

// dQ of the dS route on the bf16x3 kernel (CODA_ATTN_DQ_X3=0: the fp32-MFMA GEMM, A/B); its K^T pieces live behind the
// dS workspace (coda_mha_bwd_ws_bytes accounts for them)
bool dq_x3_route(int s_len) {
  static const bool on = [] { const char *e = getenv("CODA_ATTN_DQ_X3"); return !e || atoi(e) != 0; }();
  return on && s_len % 64 == 0;
}
size_t dq_x3_ws_bytes(int b, int h, int s_len) {
  return dq_x3_route(s_len) ? static_cast<size_t>(b) * h * (s_len / 64) * kDqChunkBytes : 0;
}

template <int D, bool GEN>
int launch_bwd_g(const MhaBwdParams &p, hipStream_t s) {
  constexpr size_t kTileBytes = sizeof(float) * 2 * kTile * (D + 4);  // one Q + one dO (or K + V) tile
  constexpr size_t kRowBytes = sizeof(float) * 2 * kTile;                // lse + delta of a tile
  constexpr int SW = (8 * kTileBytes <= 160 * 1024) ? 8 : 4;
  // double-buffered Q / dO staging pays on long query sequences (encoder: -5 %); with 256 queries
  // (decoder memory) there are only 8 stages and the prologue eats the gain
  // dS workspace route (long sequences, no mask, ragged tails excluded): dK/dV kernel writes dS, dQ is one GEMM
  // (The decoder's shapes -- 256 queries -- were tried too, with the keys split over the waves of a 32-query workgroup:
  // 0.092 ms against the two-kernel form's 0.091 for 256 x 2048, the dK/dV kernel 8 % slower: long sequences only.)
  const bool via_ds = D == 64 && !GEN && p.ds != nullptr && p.s >= 1024 && p.l >= 1024 && (p.parts & 6) == 6 &&
                      !p.fuse_delta && double_buffered();
  auto run_dkv = [&]() -> int {
  if (!(p.parts & 2)) return CODA_OK;
  KernelTimer timer(2, p.l, p.s, s);
  // S and dP on the bf16 matrix cores at fp32 accuracy (mha_bwd_dkv_x3_kernel; CODA_ATTN_DKV_X3=0: all four products
  // on the fp32 MFMA, A/B)
  static const bool dkv_x3 = [] { const char *e = getenv("CODA_ATTN_DKV_X3"); return !e || atoi(e) != 0; }();
  if (D == 64 && !GEN && dkv_x3 && p.s >= 1024 && p.l >= 1024 && double_buffered()) {
    if (via_ds) {
      auto kern = mha_bwd_dkv_x3_kernel<4, true>;
      int st = set_lds(kern, kDkvX3Lds);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.s, kTile * 4), p.b * p.h), dim3(256), kDkvX3Lds, s, p);
    } else {
      auto kern = mha_bwd_dkv_x3_kernel<4, false>;
      int st = set_lds(kern, kDkvX3Lds);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.s, kTile * 4), p.b * p.h), dim3(256), kDkvX3Lds, s, p);
    }
  } else if (via_ds) {
    auto kern = mha_bwd_dkv_kernel<D, 4, false, false, true, D == 64>;
    const size_t lds = 2 * (kTileBytes + kRowBytes);
    int st = set_lds(kern, lds);
    if (st != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.s, kTile * 4), p.b * p.h), dim3(256), lds, s, p);
  } else if (p.s >= 1024 && p.l >= 1024 && double_buffered()) {
    auto kern = mha_bwd_dkv_kernel<D, 4, false, GEN, true>;
    const size_t lds = 2 * (kTileBytes + kRowBytes);  // two single-tile stages
    int st = set_lds(kern, lds);
    if (st != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.s, kTile * 4), p.b * p.h), dim3(256), lds, s, p);
  } else if (p.s >= 1024) {
    auto kern = mha_bwd_dkv_kernel<D, 4, false, GEN>;
    const size_t lds = 2 * (kTileBytes + kRowBytes);
    int st = set_lds(kern, lds);
    if (st != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.s, kTile * 4), p.b * p.h), dim3(256), lds, s, p);
  } else {
    auto kern = mha_bwd_dkv_kernel<D, 4, true, GEN>;
    const size_t lds = 4 * (kTileBytes + kRowBytes);
    int st = set_lds(kern, lds);
    if (st != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.s, kTile), p.b * p.h), dim3(256), lds, s, p);
  }
  return CODA_OK;
  };
  auto run_dq = [&]() -> int {
  if (!(p.parts & 4)) return CODA_OK;
  if constexpr (D == 64) {
    if (via_ds && dq_x3_route(p.s)) {  // bf16x3: K^T pieces behind the dS workspace, then the GEMM as an HBM stream
      unsigned char *ktp = reinterpret_cast<unsigned char *>(p.ds + static_cast<size_t>(p.b) * p.h * p.l * p.s);
      {
        KernelTimer timer(7, p.l, p.s, s);  // kind 7: K^T pieces
        mha_launch(mha_kt_pieces_kernel, dim3(p.s / 64, p.b * p.h), dim3(256), 0, s, p, reinterpret_cast<__bf16 *>(ktp));
      }
      KernelTimer timer(4, p.l, p.s, s);
      auto kern = mha_bwd_dq_x3_kernel;
      int st = set_lds(kern, 2 * kDqLdsBuf);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, kTile * 4), p.b * p.h), dim3(256), 2 * kDqLdsBuf, s, p,
                 static_cast<const unsigned char *>(ktp));
      return CODA_OK;
    }
    if (via_ds) {
      KernelTimer timer(4, p.l, p.s, s);  // kind 4: the dS K GEMM (2 L S d flops per head)
      // 64-key chunks: two workgroups per CU (181 registers); 128-key chunks with one measured slower (0.31 vs 0.25 ms)
      constexpr size_t glds = sizeof(float) * 2 * 64 * D;  // two K chunks
      auto kern = mha_bwd_dq_gemm_kernel<D, 64>;
      int st = set_lds(kern, glds);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, kTile * 4), p.b * p.h), dim3(256), glds, s, p);
      return CODA_OK;
    }
  }
  KernelTimer timer(3, p.l, p.s, s);
  if (p.l >= 1024 && double_buffered()) {
    auto kern = mha_bwd_dq_kernel<D, 4, false, GEN, true>;
    int st = set_lds(kern, 4 * kTileBytes);
    if (st != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile * 4), p.b * p.h), dim3(256), 4 * kTileBytes, s, p);
  } else if (p.l >= 1024) {
    auto kern = mha_bwd_dq_kernel<D, 4, false, GEN>;
    int st = set_lds(kern, 4 * kTileBytes);
    if (st != CODA_OK) return st;
    mha_launch(kern, dim3(ceil_div(p.l, kTile * 4), p.b * p.h), dim3(256), 4 * kTileBytes, s, p);
  } else {
    // K / V fragments straight from L2 into registers (mha_bwd_dq_direct_kernel): head width 128 only -- 250 -> 133 us
    // on 256 x 2048; at width 64 the LDS-staged kernel below is the faster one (83 vs 93 us: 32 narrow VMEM
    // instructions per tile against 16 wide ones + LDS reads).  CODA_ATTN_DIRECT_DQ=1 forces it (A/B)
    static const bool force_dq = [] { const char *e = getenv("CODA_ATTN_DIRECT_DQ"); return e && atoi(e) != 0; }();
    const int share = D == 64 && p.s >= 1024 ? split_query_tiles(ceil_div(p.l, 2 * kTile) * p.b * p.h) : 0;
    if (share == 1) {  // K / V staged once for two query tiles (see mha_fwd_kernel)
      auto kern = mha_bwd_dq_kernel<D, 4, true, GEN, true, 2>;
      int st = set_lds(kern, 8 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, 2 * kTile), p.b * p.h), dim3(8 * kWave), 8 * kTileBytes, s, p);
    } else if (share == 2 && p.s % (2 * 4 * kTile) == 0) {  // ... and the keys in two halves: dQ accumulated
      const size_t row_bytes = sizeof(float) * p.h * D, rows = static_cast<size_t>(p.l) * p.b;
      const hipError_t e = p.lddq == p.h * D ? hipMemsetAsync(p.dq, 0, row_bytes * rows, s)
                                             : hipMemset2DAsync(p.dq, sizeof(float) * p.lddq, 0, row_bytes, rows, s);
      if (e != hipSuccess) return static_cast<int>(e);
      auto kern = mha_bwd_dq_kernel<D, 4, true, GEN, true, 2, 2>;
      int st = set_lds(kern, 8 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(2 * ceil_div(p.l, 2 * kTile), p.b * p.h), dim3(8 * kWave), 8 * kTileBytes, s, p);
    } else if (split_direct() && (D > 64 || force_dq)) {
      constexpr size_t mlds = sizeof(float) * 3 * (D / 32 * 16) * kWave;
      auto kern = mha_bwd_dq_direct_kernel<D, 4, GEN>;
      int st = set_lds(kern, mlds);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(4 * kWave), mlds, s, p);
    } else if (split_double_buffered() && 8 * kTileBytes <= 160 * 1024) {
      auto kern = mha_bwd_dq_kernel<D, 4, true, GEN, true>;
      int st = set_lds(kern, 8 * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(4 * kWave), 8 * kTileBytes, s, p);
    } else {
      auto kern = mha_bwd_dq_kernel<D, SW, true, GEN>;
      int st = set_lds(kern, SW * kTileBytes);
      if (st != CODA_OK) return st;
      mha_launch(kern, dim3(ceil_div(p.l, kTile), p.b * p.h), dim3(SW * kWave), SW * kTileBytes, s, p);
    }
  }
  return CODA_OK;
  };
  // fuse_delta: the dQ kernel writes delta, so it runs first
  int st = p.fuse_delta ? run_dq() : run_dkv();
  if (st != CODA_OK) return st;
  st = p.fuse_delta ? run_dkv() : run_dq();
  return st != CODA_OK ? st : launch_status();
}

// problems the one-kernel backward takes: head width 64, whole tiles, at most 1024 queries against at least 1024 keys
bool fused_bwd_takes(int b, int h, int l, int s, int d) {
  return b > 0 && h > 0 && d == 64 && l >= kTile && l < 1024 && l % kTile == 0 && s >= 1024 && s % (4 * kTile) == 0;
}
// ... and its short-key-sequence form (mha_bwd_fused_short_kernel): whole tiles, both sequences below 1024
bool fused_short_bwd_takes(int b, int h, int l, int s, int d) {
  return b > 0 && h > 0 && d == 64 && l >= kTile && l < 1024 && l % kTile == 0 && s >= kTile && s < 1024 && s % kTile == 0;
}
size_t fused_bwd_ws_bytes(int b, int h, int l, int s, bool short_keys) {  // partial dQ tiles: one (l x 64) per key block
  return sizeof(float) * static_cast<size_t>(b) * h * (s / (short_keys ? kTile : 4 * kTile)) * l * 64;
}

template <int D>
int launch_bwd(const MhaBwdParams &p, hipStream_t s) {
  clear_sticky_error();
  const size_t nrows = static_cast<size_t>(p.l) * p.b * p.h;
  // The whole backward with the fp32-MFMA kernels: delta is formed inside the dQ kernel (CODA_ATTN_FUSE_DELTA=0: A/B)
  static const bool fuse_ok = [] { const char *e = getenv("CODA_ATTN_FUSE_DELTA"); return !e || atoi(e) != 0; }();
  const bool ds_route = D == 64 && p.ds != nullptr && mfma_dtype() == 0 && p.mask == nullptr && p.l >= 1024 && p.s >= 1024 &&
                        p.l % kTile == 0 && p.s % kTile == 0 && (p.parts & 6) == 6;
  // one kernel for dK, dV and the partial dQ tiles + a reduction (mha_bwd_fused_kernel): short query sequences against
  // long key sequences -- the decoder's cross-attention (CODA_ATTN_FUSED_BWD=0: the two-kernel form, A/B)
  static const bool fused_ok = [] { const char *e = getenv("CODA_ATTN_FUSED_BWD"); return !e || atoi(e) != 0; }();
  const bool fused_any = fused_ok && !ds_route && p.ds != nullptr && mfma_dtype() == 0 && p.mask == nullptr;
  const bool fused_route = fused_any && fused_bwd_takes(p.b, p.h, p.l, p.s, D) && (p.parts & 6) == 6;
  const bool fused_short = fused_any && fused_short_bwd_takes(p.b, p.h, p.l, p.s, D) && (p.parts & 7) == 7;
  const bool fuse = fuse_ok && mfma_dtype() == 0 && (p.parts & 7) == 7 && !ds_route && !fused_route && !fused_short;
  if ((p.parts & 1) && !fuse && !fused_short && !fused_route) {  // (the one-kernel forms sum dO * O themselves)
    KernelTimer timer(1, p.l, p.s, s);
    mha_launch((mha_delta_kernel<D>), dim3(static_cast<unsigned>((nrows + 255) / 256)), dim3(256), 0, s, p);
  }
  if (mfma_dtype() == 1) {
    int st = CODA_OK;
    if (p.parts & 2) {
      KernelTimer timer(2, p.l, p.s, s);
      st = mha_bwd_dkv_bf16(p, D, s);
    }
    if (st != CODA_OK) return st;
    if (p.parts & 4) {
      KernelTimer timer(3, p.l, p.s, s);
      st = mha_bwd_dq_bf16(p, D, s);
    }
    return st != CODA_OK ? st : launch_status();
  }
  MhaBwdParams rest = p;  // what the fp32-MFMA kernels still have to do
  rest.parts = p.parts & ~1;
  rest.fuse_delta = fuse ? 1 : 0;
  if (!ds_route) rest.ds = nullptr;
  if (mfma_dtype() == 2) {
    int st = CODA_OK;
    if ((p.parts & 2) && mha_x3_takes_dkv(p, D)) {
      KernelTimer timer(2, p.l, p.s, s);
      st = mha_bwd_dkv_x3(p, D, s);
      rest.parts &= ~2;
    }
    if (st != CODA_OK) return st;
    if ((p.parts & 4) && mha_x3_takes_dq(p, D)) {
      KernelTimer timer(3, p.l, p.s, s);
      st = mha_bwd_dq_x3(p, D, s);
      rest.parts &= ~4;
    }
    if (st != CODA_OK) return st;
  }
  if (!(rest.parts & 6)) return launch_status();
  if constexpr (D == 64) {
    if (fused_route) {
      constexpr int KW = 4;
      constexpr size_t lds = sizeof(float) * (2 * kTile * (D + 4) + 2 * kTile + (D + kTile) * (KW * kTile + 4));
      const int nkb = p.s / (KW * kTile);
      {
        KernelTimer timer(5, p.l, p.s, s);  // kind 5: dK, dV and the partial dQ tiles (10 L S d flops per head)
        auto kern = mha_bwd_fused_kernel<KW>;
        int st = set_lds(kern, lds);
        if (st != CODA_OK) return st;
        mha_launch(kern, dim3(nkb, p.b * p.h), dim3(KW * kWave), lds, s, p);
      }
      {
        KernelTimer timer(6, p.l, p.s, s);  // kind 6: the key blocks' partial dQ tiles summed in order
        mha_launch(mha_dq_reduce_kernel<0>, dim3(p.l / kTile, p.b * p.h), dim3(256), 0, s, p, nkb);
      }
      return launch_status();
    }
    if (fused_short) {  // delta is formed inside
      constexpr int KW = 8;
      constexpr size_t lds = sizeof(float) * (2 * KW * kTile * (D + 4) + 2 * KW * kTile + kTile * (D + 4));
      const int nkb = p.s / kTile;
      {
        KernelTimer timer(5, p.l, p.s, s);
        auto kern = mha_bwd_fused_short_kernel<KW>;
        int st = set_lds(kern, lds);
        if (st != CODA_OK) return st;
        mha_launch(kern, dim3(nkb, p.b * p.h), dim3(KW * kWave), lds, s, p);
      }
      {
        KernelTimer timer(6, p.l, p.s, s);
        mha_launch(mha_dq_reduce_kernel<1>, dim3(p.l / kTile, p.b * p.h), dim3(256), 0, s, p, nkb);
      }
      return launch_status();
    }
  }
  const bool gen = p.mask != nullptr || (p.l % kTile) != 0 || (p.s % kTile) != 0;
  return gen ? launch_bwd_g<D, true>(rest, s) : launch_bwd_g<D, false>(rest, s);
}

}  // namespace
}  // namespace coda

CODA_API int coda_mha_fwd_f32(const float *q, const float *k, const float *v, const uint8_t *mask,
                              float *out, float *lse, int b, int h, int l, int s, int d, int ldq, int ldk,
                              int ldv, float scale, float dropout_p, uint64_t seed,
                              const uint64_t *seed_dev, void *stream) {
  using namespace coda;
  if (b < 0 || h <= 0 || l < 0 || s < 0 || (d != 64 && d != 128) || dropout_p < 0.f || dropout_p >= 1.f ||
      ldq < h * d || ldk < h * d || ldv < h * d || (ldq | ldk | ldv) % 4 != 0)
    return CODA_EINVAL;
  if (b == 0 || l == 0) return CODA_OK;
  if (!q || !out || !lse || (s > 0 && (!k || !v))) return CODA_EINVAL;
  MhaParams p;
  p.q = q; p.k = k; p.v = v; p.mask = mask; p.out = out; p.lse = lse;
  p.b = b; p.h = h; p.l = l; p.s = s;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.scale = scale;
  p.thresh16 = drop_threshold(dropout_p);
  p.inv_keep = 1.0f / (1.0f - dropout_p);
  p.seed = static_cast<uint32_t>(seed ^ (seed >> 32));
  p.seed_dev = seed_dev;
  p.xcd_map = xcd_mapped();
  hipStream_t st = static_cast<hipStream_t>(stream);
  return d == 64 ? launch_fwd<64>(p, st) : launch_fwd<128>(p, st);
}

CODA_API int coda_mha_bwd_f32(const float *q, const float *k, const float *v, const uint8_t *mask,
                              const float *out, const float *lse, const float *dout, float *dq,
                              float *dk, float *dv, float *delta, int b, int h, int l, int s, int d,
                              int ldq, int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                              float dropout_p, uint64_t seed, const uint64_t *seed_dev, void *stream) {
  return coda_mha_bwd_parts_f32(q, k, v, mask, out, lse, dout, dq, dk, dv, delta, b, h, l, s, d, ldq, ldk, ldv, lddq,
                                lddk, lddv, scale, dropout_p, seed, seed_dev, 7, stream);
}

CODA_API int coda_mha_bwd_parts_f32(const float *q, const float *k, const float *v, const uint8_t *mask,
                                    const float *out, const float *lse, const float *dout, float *dq,
                                    float *dk, float *dv, float *delta, int b, int h, int l, int s, int d,
                                    int ldq, int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                                    float dropout_p, uint64_t seed, const uint64_t *seed_dev, int parts,
                                    void *stream) {
  using namespace coda;
  if (parts <= 0 || parts > 7) return CODA_EINVAL;
  if (lddq == 0) lddq = h * d;
  if (lddk == 0) lddk = h * d;
  if (lddv == 0) lddv = h * d;
  if (b < 0 || h <= 0 || l < 0 || s < 0 || (d != 64 && d != 128) || dropout_p < 0.f || dropout_p >= 1.f ||
      ldq < h * d || ldk < h * d || ldv < h * d || (ldq | ldk | ldv) % 4 != 0 || lddq < h * d || lddk < h * d ||
      lddv < h * d || (lddq | lddk | lddv) % 4 != 0)
    return CODA_EINVAL;
  if (b == 0 || (l == 0 && s == 0)) return CODA_OK;
  if (l == 0 || s == 0) return CODA_EINVAL;
  if (!q || !k || !v || !out || !lse || !dout || !delta) return CODA_EINVAL;
  if (((parts & 4) && !dq) || ((parts & 2) && (!dk || !dv))) return CODA_EINVAL;
  MhaBwdParams p;
  p.parts = parts;
  p.q = q; p.k = k; p.v = v; p.out = out; p.lse = lse; p.dout = dout; p.mask = mask;
  p.dq = dq; p.dk = dk; p.dv = dv; p.delta = delta;
  p.b = b; p.h = h; p.l = l; p.s = s;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.scale = scale;
  p.thresh16 = drop_threshold(dropout_p);
  p.inv_keep = 1.0f / (1.0f - dropout_p);
  p.seed = static_cast<uint32_t>(seed ^ (seed >> 32));
  p.seed_dev = seed_dev;
  p.xcd_map = xcd_mapped();
  {  // optional dS workspace of coda_mha_bwd_ws_f32 (per-call options, common.hip.h)
    const CallOptions &o = call_options();
    const size_t need = coda_mha_bwd_ws_bytes(b, h, l, s, d);  // 0: this problem uses none
    if (need && o.attn_ds_ws && o.attn_ds_bytes >= need && (reinterpret_cast<uintptr_t>(o.attn_ds_ws) & 15) == 0)
      p.ds = static_cast<float *>(o.attn_ds_ws);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return d == 64 ? launch_bwd<64>(p, st) : launch_bwd<128>(p, st);
}

CODA_API size_t coda_mha_bwd_ws_bytes(int b, int h, int l, int s, int d) {
  // the dS route applies to long unmasked sequences at head width 64 (the encoder's self-attention); 0 = not used
  // ... and the one-kernel backward of short query sequences against long key sequences (the decoder's cross-attention)
  // keeps the key blocks' partial dQ tiles there
  if (coda::fused_bwd_takes(b, h, l, s, d)) return coda::fused_bwd_ws_bytes(b, h, l, s, false);
  if (coda::fused_short_bwd_takes(b, h, l, s, d)) return coda::fused_bwd_ws_bytes(b, h, l, s, true);
  if (b <= 0 || h <= 0 || d != 64 || l < 1024 || s < 1024 || l % 32 != 0 || s % 32 != 0) return 0;
  return sizeof(float) * static_cast<size_t>(b) * h * l * s + coda::dq_x3_ws_bytes(b, h, s);  // dS + the K^T pieces of dQ
}

CODA_API int coda_mha_bwd_ws_f32(const float *q, const float *k, const float *v, const uint8_t *mask, const float *out,
                                 const float *lse, const float *dout, float *dq, float *dk, float *dv, float *delta, int b,
                                 int h, int l, int s, int d, int ldq, int ldk, int ldv, int lddq, int lddk, int lddv,
                                 float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev, void *workspace,
                                 size_t workspace_bytes, int mfma_dtype, void *stream) {
  if (mfma_dtype < -1 || mfma_dtype > 2) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.mfma_dtype = mfma_dtype;
  o.attn_ds_ws = workspace;
  o.attn_ds_bytes = workspace ? workspace_bytes : 0;
  coda::ScopedCallOptions scope(o);
  return coda_mha_bwd_parts_f32(q, k, v, mask, out, lse, dout, dq, dk, dv, delta, b, h, l, s, d, ldq, ldk, ldv, lddq, lddk,
                                lddv, scale, dropout_p, seed, seed_dev, 7, stream);
}

CODA_API int coda_mha_fwd_opt_f32(const float *q, const float *k, const float *v, const uint8_t *mask, float *out,
                                  float *lse, int b, int h, int l, int s, int d, int ldq, int ldk, int ldv, float scale,
                                  float dropout_p, uint64_t seed, const uint64_t *seed_dev, int mfma_dtype, void *stream) {
  if (mfma_dtype < -1 || mfma_dtype > 2) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.mfma_dtype = mfma_dtype;
  coda::ScopedCallOptions scope(o);
  return coda_mha_fwd_f32(q, k, v, mask, out, lse, b, h, l, s, d, ldq, ldk, ldv, scale, dropout_p, seed, seed_dev, stream);
}

CODA_API int coda_mha_bwd_parts_opt_f32(const float *q, const float *k, const float *v, const uint8_t *mask,
                                        const float *out, const float *lse, const float *dout, float *dq, float *dk,
                                        float *dv, float *delta, int b, int h, int l, int s, int d, int ldq, int ldk,
                                        int ldv, int lddq, int lddk, int lddv, float scale, float dropout_p,
                                        uint64_t seed, const uint64_t *seed_dev, int parts, int mfma_dtype, void *stream) {
  if (mfma_dtype < -1 || mfma_dtype > 2) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.mfma_dtype = mfma_dtype;
  coda::ScopedCallOptions scope(o);
  return coda_mha_bwd_parts_f32(q, k, v, mask, out, lse, dout, dq, dk, dv, delta, b, h, l, s, d, ldq, ldk, ldv, lddq, lddk,
                                lddv, scale, dropout_p, seed, seed_dev, parts, stream);
}

CODA_API int coda_mha_get_mfma_dtype(void) { return coda::default_mfma_dtype(); }

CODA_API int coda_mha_timing_enable(int min_len) {
  using namespace coda;
  timing_clear();
  g_timing_min_len = min_len;
  g_timing_kinds = ~0u;
  return CODA_OK;
}

CODA_API int coda_mha_timing_enable_kinds(int min_len, unsigned kind_mask) {
  using namespace coda;
  timing_clear();
  g_timing_min_len = min_len;
  g_timing_kinds = kind_mask;
  return CODA_OK;
}

CODA_API int coda_mha_timing_collect(int *kind, int *l, int *s, float *ms, int cap) {
  using namespace coda;
  if (cap < 0 || (cap > 0 && (!kind || !l || !s || !ms))) return CODA_EINVAL;
  int n = 0;
  for (const auto &r : g_timing) {
    if (n >= cap) break;
    hipError_t e = hipEventSynchronize(r.e1);
    if (e != hipSuccess) return -(1000 + static_cast<int>(e));
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.e0, r.e1);
    if (e != hipSuccess) return -(1000 + static_cast<int>(e));
    kind[n] = r.kind; l[n] = r.l; s[n] = r.s; ms[n] = t;
    ++n;
  }
  return n;
}
