// fps.hip -- furthest point sampling for gfx950 (MI355X).
//
// Replaces the reference's one-512-thread-block-per-scene CUDA kernel
// (third_party_pointnet2/pointnet2/_ext_src/src/sampling_gpu.cu:72-232).
//
// Design (MI355X-first):
//  * FPS is m-1 strictly dependent arg-max rounds; the bound is the latency of
//    one round, not HBM.  One workgroup (up to 16 wave64s = one CU) owns a
//    scene and keeps the WHOLE cloud plus the running min-distances in VGPRs
//    (P points per thread) for all rounds: HBM is read once (12 N bytes/scene),
//    written once (4 m bytes/scene).
//  * One s_barrier per round: wave-level arg-max with DPP row shifts /
//    row broadcasts on a 64-bit composite key, 16 wave results exchanged
//    through a double-buffered LDS slot, re-reduced redundantly by every wave.
//  * The composite key (dist bits, ~tie-rank, index) makes the arg-max
//    independent of the reduction topology while reproducing the reference's
//    tie-break exactly: the reference's 2^k-thread strided scan keeps the
//    lowest k per thread (strict '>', :111-112) and its LDS tree keeps slot
//    idx1 on ties (:60-68), so among equal distances the winner is the point
//    with the smallest bit-reversed (k mod T), then the smallest k, where
//    T = min(512, 2^floor(log2 N)) (include/cuda_utils.h:17-21).
#include "common.hip.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace coda {
namespace {

// `(double)mag <= 1e-3` (sampling_gpu.cu:104) for a float mag is `mag <= T`
// with T the largest float not above the double 0.001: 1e-3f rounds UP to
// 0x3A83126F (0.00100000005), so T = 0x3A83126E.
template <int DM>
__device__ __forceinline__ bool fps_skipped(float x, float y, float z) {
  const float mag = sqdist3<DM>(x, y, z);
  return mag <= __uint_as_float(0x3A83126Eu);
}

__device__ __forceinline__ uint32_t bitrev_low(uint32_t v, int bits) {
  return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

// Tie rank of point k (smaller wins among equal distances).
__device__ __forceinline__ uint32_t tie_rank(uint32_t k, int log2T) {
  const uint32_t kmod = k & ((1u << log2T) - 1u);
  return (bitrev_low(kmod, log2T) << 23) | (k >> log2T);
}
__device__ __forceinline__ uint32_t rank_to_index(uint32_t r, int log2T) {
  const uint32_t kdiv = r & ((1u << 23) - 1u);
  const uint32_t kmod = bitrev_low(r >> 23, log2T);
  return (kdiv << log2T) | kmod;
}

// One DPP step of a 64-bit unsigned max: partner value comes from the lane
// selected by CTRL; lanes without a valid partner see key 0 (the identity).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void kmax_step(uint32_t &hi, uint32_t &lo) {
  const uint32_t ohi = __builtin_amdgcn_update_dpp(0u, hi, CTRL, ROW_MASK, 0xf, false);
  const uint32_t olo = __builtin_amdgcn_update_dpp(0u, lo, CTRL, ROW_MASK, 0xf, false);
  const uint64_t mine = (static_cast<uint64_t>(hi) << 32) | lo;
  const uint64_t other = (static_cast<uint64_t>(ohi) << 32) | olo;
  if (other > mine) {
    hi = ohi;
    lo = olo;
  }
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// Max over the 64 lanes; result valid in lane 63 and returned wave-uniform.
__device__ __forceinline__ void wave_kmax(uint32_t &hi, uint32_t &lo) {
  kmax_step<DPP_ROW_SHR1, 0xf>(hi, lo);
  kmax_step<DPP_ROW_SHR2, 0xf>(hi, lo);
  kmax_step<DPP_ROW_SHR4, 0xf>(hi, lo);
  kmax_step<DPP_ROW_SHR8, 0xf>(hi, lo);
  kmax_step<DPP_ROW_BCAST15, 0xa>(hi, lo);
  kmax_step<DPP_ROW_BCAST31, 0xc>(hi, lo);
  hi = __builtin_amdgcn_readlane(hi, 63);
  lo = __builtin_amdgcn_readlane(lo, 63);
}

// Max over the first NW lanes of a 16-lane row (NW power of two <= 16);
// returned wave-uniform.
template <int NW>
__device__ __forceinline__ void row_kmax(uint32_t &hi, uint32_t &lo) {
  if (NW > 1) kmax_step<DPP_ROW_SHR1, 0xf>(hi, lo);
  if (NW > 2) kmax_step<DPP_ROW_SHR2, 0xf>(hi, lo);
  if (NW > 4) kmax_step<DPP_ROW_SHR4, 0xf>(hi, lo);
  if (NW > 8) kmax_step<DPP_ROW_SHR8, 0xf>(hi, lo);
  hi = __builtin_amdgcn_readlane(hi, NW - 1);
  lo = __builtin_amdgcn_readlane(lo, NW - 1);
}

// Exchange the per-wave keys through LDS (slot parity = round parity; one
// barrier per round is enough, see the header comment) and reduce them.
template <int NW>
__device__ __forceinline__ uint32_t block_argmax(uint32_t hi, uint32_t lo, uint2 (*s_key)[NW],
                                                 int parity, int log2T) {
  wave_kmax(hi, lo);
  if (NW > 1) {
    if (lane_id() == 0) s_key[parity][wave_id()] = make_uint2(hi, lo);
    __syncthreads();
    const uint2 kk = s_key[parity][lane_id() & (NW - 1)];
    hi = kk.x;
    lo = kk.y;
    row_kmax<NW>(hi, lo);
  }
  // hi == 0: no thread saw a participating point -> reference yields besti = 0.
  return hi == 0u ? 0u : rank_to_index(~lo, log2T);
}

// Selected indices are staged in LDS and written out 256 at a time: __syncthreads() waits for
// ALL outstanding memory operations of a wave (s_waitcnt vmcnt(0)), so a global store per round
// would put a full HBM write latency (~0.5 us) on the critical path of every round.
constexpr int kOutRing = 256;
__device__ __forceinline__ void out_put(int32_t *s_out, int j, int32_t v) {
  if (threadIdx.x == 0) s_out[j & (kOutRing - 1)] = v;
}
// called by ALL threads, after out_put of round j (j, m block-uniform)
__device__ __forceinline__ void out_flush(const int32_t *s_out, int j, int m, int32_t *__restrict__ out) {
  if ((j & (kOutRing - 1)) == kOutRing - 1 || j == m - 1) {
    __syncthreads();
    const int base = j & ~(kOutRing - 1);
    if (static_cast<int>(threadIdx.x) <= (j & (kOutRing - 1))) out[base + threadIdx.x] = s_out[threadIdx.x];
  }
}

// ---- register-resident kernel: P points per thread, THREADS per scene --------
template <int P, int THREADS, int DM>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(const float *__restrict__ xyz, int n,
                                                          int m, int log2T,
                                                          int32_t *__restrict__ idx) {
  constexpr int NW = THREADS / kWave;
  __shared__ uint2 s_key[2][NW];
  __shared__ int32_t s_out[kOutRing];

  const int tid = threadIdx.x;
  const float *__restrict__ pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
  int32_t *__restrict__ out = idx + static_cast<size_t>(blockIdx.x) * m;

  float x[P], y[P], z[P], t[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = tid + i * THREADS;
    if (k < n) {
      x[i] = pts[k * 3 + 0];
      y[i] = pts[k * 3 + 1];
      z[i] = pts[k * 3 + 2];
      // running min distance: 1e10 (sampling.cpp:75-77); -1 marks a point that
      // never takes part (skip rule :103-104): fminf(d, -1) stays -1 and
      // `-1 > best` is false for best >= -1.
      t[i] = fps_skipped<DM>(x[i], y[i], z[i]) ? -1.0f : 1e10f;
    } else {
      x[i] = y[i] = z[i] = 0.0f;
      t[i] = -1.0f;
    }
  }

  out_put(s_out, 0, 0);  // :88-89
  out_flush(s_out, 0, m, out);
  float cx = pts[0], cy = pts[1], cz = pts[2];

  for (int j = 1; j < m; ++j) {
    float best = -1.0f;  // :94
    int besti = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float d = sqdist3<DM>(__fsub_rn(x[i], cx), __fsub_rn(y[i], cy), __fsub_rn(z[i], cz));
      const float d2 = fminf(d, t[i]);  // :109
      t[i] = d2;                        // :110
      // strict '>' in ascending i: thread-local ties keep the lowest k, and
      // all k of a thread share (k mod T) because T divides THREADS.
      if (d2 > best) {
        best = d2;
        besti = i;
      }
    }
    uint32_t hi = 0u, lo = 0u;
    if (best >= 0.0f) {
      hi = __float_as_uint(best) + 1u;  // non-negative floats order like uints
      lo = ~tie_rank(static_cast<uint32_t>(tid + besti * THREADS), log2T);
    }
    const uint32_t old = block_argmax<NW>(hi, lo, s_key, j & 1, log2T);
    out_put(s_out, j, static_cast<int32_t>(old));  // :173-174
    out_flush(s_out, j, m, out);
    cx = pts[old * 3 + 0];  // wave-uniform address -> scalar loads
    cy = pts[old * 3 + 1];
    cz = pts[old * 3 + 2];
  }
}

// ---- v2: 512 threads == T (the reference block size), packed pairs, 32-bit DPP ------
//
// With THREADS == T == 512 every thread owns exactly one residue (k mod T = tid), so
// the cross-thread tie order is bitrev9(tid) alone and the arg-max splits into two
// cheap 32-bit wave reductions (float max, then u32 min of bitrev9(tid) among the
// lanes that hold the max) done with DPP-fused VALU ops.  The per-point update is
// written on float2 pairs (v_pk_add/v_pk_mul, still one rounding per operation), the
// running minimum uses a bare v_min_f32 and the thread maximum v_max3_f32 (no
// canonicalisation moves), and the slot index of the maximum is recovered with an
// equality chain instead of being dragged through the update loop.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// Wave-wide max of a float; valid in lane 63 (returned uniform).  DPP needs two wait
// states after the VALU write of its source, hence the s_nop 1 between the steps.
__device__ __forceinline__ float wave_max_f32(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// LX: the cloud is also kept in LDS (n * 12 bytes, dynamic) so that the winner's coordinates -- the only thing a round
// reads back -- come from there: three dependent L2 round trips (~650 clocks of a ~2400-clock round, tools/fps_prof.py)
// become one LDS read.  Used for small clouds (the object queries' sampling 2048 -> 256).
template <int P, int DM, bool LX = false>  // points per thread (even); 512 threads, n in [512, 512*P]
__global__ __launch_bounds__(512) void fps_t512_kernel(const float *__restrict__ xyz, int n, int m,
                                                       int32_t *__restrict__ idx) {
  constexpr int THREADS = 512, H = P / 2;
  __shared__ unsigned long long s_slot[3];
  __shared__ int32_t s_out[kOutRing];
  extern __shared__ float s_xyz[];  // LX: [n][3]

  const int tid = threadIdx.x;
  const int lane = lane_id();
  const float *__restrict__ pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
  int32_t *__restrict__ out = idx + static_cast<size_t>(blockIdx.x) * m;
  const uint32_t brev = __brev(static_cast<uint32_t>(tid)) >> 23;  // bitrev9(tid)

  f32x2 x[H], y[H], z[H];
  float t[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = tid + i * THREADS;
    float px = 0.0f, py = 0.0f, pz = 0.0f, pt = -1.0f;
    if (k < n) {
      px = pts[k * 3 + 0];
      py = pts[k * 3 + 1];
      pz = pts[k * 3 + 2];
      pt = fps_skipped<DM>(px, py, pz) ? -1.0f : 1e10f;  // see fps_reg_kernel
    }
    x[i / 2][i % 2] = px;
    y[i / 2][i % 2] = py;
    z[i / 2][i % 2] = pz;
    t[i] = pt;
  }
  if (tid < 3) s_slot[tid] = 0ull;
  out_put(s_out, 0, 0);
  float cx = pts[0], cy = pts[1], cz = pts[2];
  if (LX) {
    for (int i = tid; i < n * 3; i += THREADS) s_xyz[i] = pts[i];
  }
  __syncthreads();
  out_flush(s_out, 0, m, out);

  for (int j = 1; j < m; ++j) {
    const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
    float best = -1.0f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const f32x2 dx = x[h] - cx2, dy = y[h] - cy2, dz = z[h] - cz2;
      f32x2 d;  // -ffp-contract=off: exactly the operations written here (common.hip.h, dot3)
      if constexpr (DM == 1) d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
      else if constexpr (DM == 2) d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      else d = (dx * dx + dy * dy) + dz * dz;
      t[2 * h] = vmin(d[0], t[2 * h]);
      t[2 * h + 1] = vmin(d[1], t[2 * h + 1]);
      best = vmax3(best, t[2 * h], t[2 * h + 1]);
    }
    int besti = 0;  // lowest slot holding the thread maximum (= lowest k of the thread)
#pragma unroll
    for (int i = P - 1; i >= 0; --i) besti = (t[i] == best) ? i : besti;

    const float wmax = wave_max_f32(best);
    const uint32_t cand = (best == wmax) ? brev : 0xffffu;
    const uint32_t wrank = wave_min_u32(cand);
    const uint64_t owner = __ballot(cand == wrank);
    const int owner_lane = __ffsll(static_cast<unsigned long long>(owner)) - 1;
    const uint32_t wbesti = __builtin_amdgcn_readlane(static_cast<uint32_t>(besti), owner_lane);
    if (lane == 0 && wmax >= 0.0f) {
      const unsigned long long key =
          (static_cast<unsigned long long>(__float_as_uint(wmax) + 1u) << 32) |
          static_cast<uint32_t>(~((wrank << 23) | wbesti));
      atomicMax(&s_slot[j % 3], key);
    }
    __syncthreads();
    const unsigned long long key = s_slot[j % 3];
    if (tid == 0) s_slot[(j + 2) % 3] = 0ull;  // next use is two barriers away
    const uint32_t rk = ~static_cast<uint32_t>(key);
    uint32_t old = ((rk & 0x7fffffu) << 9) | (__brev(rk >> 23) >> 23);
    if ((key >> 32) == 0ull) old = 0u;  // nothing participated: reference besti = 0
    old = __builtin_amdgcn_readfirstlane(old);
    out_put(s_out, j, static_cast<int32_t>(old));
    out_flush(s_out, j, m, out);
    const float *src = LX ? s_xyz : pts;
    cx = src[old * 3 + 0];
    cy = src[old * 3 + 1];
    cz = src[old * 3 + 2];
  }
}

template <int P>
void launch_t512(const float *xyz, int b, int n, int m, int32_t *idx, hipStream_t s) {
  const size_t lx_bytes = sizeof(float) * 3 * static_cast<size_t>(n);
  if (lx_bytes <= 48 * 1024) {  // (with the static LDS well inside the 64 KB a launch gets without opting in)
    CODA_DISPATCH_DM(distance_mode(), hipLaunchKernelGGL((fps_t512_kernel<P, DM, true>), dim3(b), dim3(512), lx_bytes, s,
                                                         xyz, n, m, idx));
    return;
  }
  CODA_DISPATCH_DM(distance_mode(),
                   hipLaunchKernelGGL((fps_t512_kernel<P, DM>), dim3(b), dim3(512), 0, s, xyz, n, m, idx));
}

bool dispatch_t512(const float *xyz, int b, int n, int m, int32_t *idx, hipStream_t s) {
  if (n < 512 || n > 512 * 40) return false;
  const int p = ceil_div(n, 512);
  if (p <= 2) launch_t512<2>(xyz, b, n, m, idx, s);
  else if (p <= 4) launch_t512<4>(xyz, b, n, m, idx, s);
  else if (p <= 8) launch_t512<8>(xyz, b, n, m, idx, s);
  else if (p <= 16) launch_t512<16>(xyz, b, n, m, idx, s);
  else if (p <= 24) launch_t512<24>(xyz, b, n, m, idx, s);
  else if (p <= 32) launch_t512<32>(xyz, b, n, m, idx, s);
  else launch_t512<40>(xyz, b, n, m, idx, s);
  return true;
}

// ---- v3: spatial buckets + exact bounding-box pruning ---------------------------------
//
// A round only changes the running distance of points that are closer to the new sample
// than to every earlier one -- after the first few rounds a small neighbourhood.  The cloud
// is therefore sorted (once, in the prologue) along a 16^3 Morton grid and cut into buckets of
// 64 consecutive points; bucket (slot j, wave w) lives in register slot j of wave w, one point
// per lane, and lane j of the wave keeps the bucket's bounding box and its current maximum running
// distance.  Per round a wave tests all its
// buckets at once (lane j: squared distance from the new sample to box j); a bucket whose box
// is no closer than its maximum running distance cannot change and is skipped:
//     d(k) >= LB(box) >= max_b temp >= temp[k]  =>  min(d(k), temp[k]) == temp[k]
// LB is evaluated with the same rounded operations as d on the clamped sample, and rounding is
// monotone, so LB <= d(k) holds bit-exactly for every k in the box.  That argument covers every
// distance mode of common.hip.h: per axis |fl(q_a - c_a)| <= |fl(p_a - c_a)| (q = clamp(c, box)),
// and each of fl(u*u), fl(u*u + t) [= fma(u, u, t)] and fl(s + t) is non-decreasing in |u|, s, t >= 0,
// so the contracted forms fma(dz,dz, fma(dx,dx, dy*dy)) are monotone in (|dx|,|dy|,|dz|) like the
// uncontracted one: the selected indices are
// IDENTICAL to the exhaustive scan's, only the work differs (measured: ~3% of the buckets are
// touched per round on room-like clouds).  The arg-max runs over the cached per-bucket maxima
// with the composite (distance, bitrev(k mod T), k) order of the kernels above; WHICH point holds a
// wave's maximum (key, coordinates) is resolved once per round and wave, and only when the bucket of
// the wave's standing candidate was touched -- running distances only decrease, so otherwise the
// candidate stands.
//
// What a round costs (tools/fps_prof.py, shader clocks, 16 waves, 8 x 20 000 -> 2048): box tests 140, bucket
// updates 390 (0.74 per wave and round, 12 per scene), candidate 85, posting the candidate 420, barrier 470 + the
// wait for the slowest wave, reading the winner back 650: 2 200 clocks = 0.99 us.  More than half is the
// exchange, and everything behind the barrier is executed by all 16 waves, four to a SIMD: an instruction
// there costs 16 clocks of the round.  Two richer exchanges were built and measured slower for that reason:
// records with round tags polled by every wave instead of the barrier (2.56 vs 2.32 ms at 8 waves), and
// accepting the runner-up in the same round when the winner provably does not change it (exact; 88 % of the
// rounds on rooms, 50 % on clouds with duplicated points: 1.83 ms on the best scene, 2.31 on the batch).
constexpr int kBucketThreads = 512, kBucketWaves = kBucketThreads / kWave;
constexpr int kMortonCells = 4096;  // 16^3

__device__ __forceinline__ uint32_t spread4(uint32_t v) {  // bits 0..3 -> bits 0,3,6,9
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ float wave_reduce_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}

struct BucketMeta {  // lane j: bucket in slot j of this wave
  float lox, loy, loz, hix, hiy, hiz;  // bounding box
  float maxt;                          // largest running distance in the bucket
};
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// SL floats per lane with WAVE-UNIFORM dynamic indexing.  The storage is one or two ext-vector register tuples
// (32 / 16 / 8 / 4 wide), which the backend indexes with s_set_gpr_idx (three instructions per access) -- a C array
// indexed through a switch is turned into a tree of flow blocks with a register copy per slot and level (measured:
// 966 clocks per bucket update, tools/fps_prof.py, of which the arithmetic is < 300).
template <int N> struct FVec { typedef float type __attribute__((ext_vector_type(N))); };
template <int SL>
struct Slots {
  static constexpr int A = SL >= 32 ? 32 : SL >= 16 ? 16 : SL >= 8 ? 8 : 4;
  static constexpr int B = SL - A;  // 0, 4, 8 or 16
  static_assert(B == 0 || B == 4 || B == 8 || B == 16, "slot counts: 4, 8, 12, 16, 20, 24, 32, 36, 40, 48");
  // a tuple of <= 8 registers is indexed with a compare / select chain per element by the backend, wider ones with
  // s_set_gpr_idx: the second tuple is 16 wide where the register budget allows it (8 waves: 256 per lane)
  static constexpr int BW = B == 0 ? 4 : (B == 8 && A == 32) ? 16 : B;
  typename FVec<A>::type a;
  typename FVec<BW>::type b;
  template <int J> __device__ __forceinline__ void put(float v) {
    if constexpr (J < A) a[J] = v;
    else b[J - A] = v;
  }
};

// One bucket (register `i` of the tuples) against the new sample: running distances, then the bucket's new maximum.
// The holder of the maximum (tie-break key, coordinates) is NOT resolved here: only the wave's best bucket needs
// it, once per round.
template <int DM, class V>
__device__ __forceinline__ float bucket_touch(const V &px, const V &py, const V &pz, V &t, int i, float cx, float cy,
                                              float cz) {
  const float d = sqdist3<DM>(__fsub_rn(px[i], cx), __fsub_rn(py[i], cy), __fsub_rn(pz[i], cz));
  float tt = t[i];
  asm volatile("v_min_f32 %0, %1, %0" : "+v"(tt) : "v"(d));  // lanes without a point keep -1
  t[i] = tt;
  return wave_max_f32(tt);
}

// The points of bucket `i` holding the distance wv: smallest key (keys are unique) and, if it beats wk, its owner.
template <class V>
__device__ __forceinline__ void bucket_resolve(const V &px, const V &py, const V &pz, const V &t, int i, uint32_t key,
                                               float wv, uint32_t &wk, float &wx, float &wy, float &wz) {
  const uint32_t cand = (t[i] == wv) ? key : 0xffffffffu;
  const uint32_t k = wave_min_u32(cand);
  if (k < wk) {
    wk = k;
    const int owner = __builtin_ctzll(__ballot(cand == k));
    wx = readlane_f(px[i], owner);
    wy = readlane_f(py[i], owner);
    wz = readlane_f(pz[i], owner);
  }
}

// Two workgroups per scene (NWG = 2, n up to 2 * 20 480): each keeps one half of the Morton-ordered buckets in its
// registers and every round the two exchange their candidates (distance, key, round tag) through a mailbox in
// global memory -- one relaxed 64-bit agent-scope atomic each way.  Two mailboxes per workgroup (round parity): a
// workgroup can only be one round ahead of its partner.  The pair is launched as blocks (scene, 0) and (scene, 1): ids s and s + b, the same XCD
// (and L2) when b is a multiple of 8.
struct FpsMailbox {
  unsigned long long kk;
  unsigned long long pad_[7];  // one 64-byte line per mailbox
};
constexpr int kFpsSpinLimit = 1 << 22;  // ~ seconds of polling: a lost partner ends the wait, not the device

// A workgroup of the pair that gave up waiting has produced WRONG indices from that round on.  That must not be
// silent (the reference's single block per scene, sampling_gpu.cu:72-176, has no such failure mode): the wave that
// gives up writes a diagnostic word (bit 31 | scene << 16 | round) to pinned host memory -- visible to the host without any synchronisation --
// and stops polling for good (one bounded wait per launch, not one per round).  Every later
// coda_furthest_point_sampling* call returns CODA_ELOST while that word is non-zero and has not been read with
// coda_fps_lost_partner_events(reset = 1).  The word is allocated at the first two-workgroup launch of the process
// (the only allocation the library ever makes on this path; mapped + portable, so every device sees it).
struct FpsStatusWord {
  unsigned int *host = nullptr;
  int err = 0;
};
const FpsStatusWord &fps_status_word(bool create) {
  static std::atomic<bool> ready{false};
  static std::mutex mu;
  static FpsStatusWord word;
  if (!ready.load(std::memory_order_acquire) && create) {
    std::lock_guard<std::mutex> g(mu);
    if (!ready.load(std::memory_order_relaxed)) {
      void *p = nullptr;
      const hipError_t e = hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent);
      if (e == hipSuccess) {
        word.host = static_cast<unsigned int *>(p);
        __atomic_store_n(word.host, 0u, __ATOMIC_RELAXED);
      } else {
        word.err = static_cast<int>(e);
        (void)hipGetLastError();
      }
      ready.store(true, std::memory_order_release);
    }
  }
  return word;  // (host == nullptr before the first two-workgroup launch: nothing can have been lost)
}
unsigned int fps_lost_events(bool reset) {
  const FpsStatusWord &w = fps_status_word(false);
  if (!w.host) return 0u;
  return reset ? __atomic_exchange_n(w.host, 0u, __ATOMIC_ACQ_REL) : __atomic_load_n(w.host, __ATOMIC_ACQUIRE);
}

// -DCODA_FPS_PROF (tools/fps_prof.py builds a private copy of this file with it): shader-clock sums of the phases
// of a round per (scene, wave), read back with coda_fps_prof_read.  Compiles to nothing in the library.
#ifdef CODA_FPS_PROF
__device__ unsigned long long g_fps_prof[64][16][8];
#define FPS_PROF_DECL unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t_ = __builtin_readcyclecounter()
#define FPS_PROF_MARK(i)                                              \
  do {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();     \
    prof_[i] += now_ - prof_t_;                                       \
    prof_t_ = now_;                                                   \
  } while (0)
#define FPS_PROF_COUNT(i, v) prof_[i] += (v)
#define FPS_PROF_STORE                                                                        \
  do {                                                                                        \
    if (lane == 0 && blockIdx.x < 64)                                                         \
      for (int q_ = 0; q_ < 8; ++q_) g_fps_prof[blockIdx.x][w + W * half][q_] = prof_[q_]; \
  } while (0)
#else
#define FPS_PROF_DECL
#define FPS_PROF_MARK(i)
#define FPS_PROF_COUNT(i, v)
#define FPS_PROF_STORE
#endif

// template parameters: SL register slots (buckets) per wave, distance mode, workgroups per scene, waves per workgroup;
// n <= 64 * W * SL * NWG
template <int SL, int DM, int NWG = 1, int W = kBucketWaves>
__global__ __launch_bounds__(64 * W) void fps_bucket_kernel(const float *__restrict__ xyz, int n, int m, int log2T,
                                                            float4 *__restrict__ sorted, int32_t *__restrict__ idx,
                                                            FpsMailbox *__restrict__ mail,
                                                            unsigned int *__restrict__ lost_events, int spin_limit,
                                                            int drop_half) {
  if constexpr (NWG == 2) {
    // test hook (coda_furthest_point_sampling_dbg_f32): the workgroup that never shows up
    if (static_cast<int>(blockIdx.y) == drop_half) return;
  }
  constexpr int TH = 64 * W;                       // threads
  constexpr int PER = kMortonCells / TH;           // histogram counters per thread in the scan
  constexpr int KB = NWG == 1 ? 15 : 16;           // bits of the point index inside the tie-break key
  constexpr int IDB = 32 - (KB + 9);               // low bits of the packed candidate left for the sender id
  static_assert(W * NWG <= (1 << IDB), "sender id bits");
  const int half = NWG == 1 ? 0 : static_cast<int>(blockIdx.y);
  __shared__ unsigned int s_hist[kMortonCells];
  __shared__ float s_red[6][W];
  __shared__ unsigned int s_wsum[W];
  __shared__ unsigned long long s_slot[3];
  __shared__ float4 s_xyz[2][W];  // per-wave candidate coordinates, round parity
  __shared__ int32_t s_out[kOutRing];
  extern __shared__ uint32_t s_keys[];  // [SL][TH]: tie-break key of the point in (slot, thread)

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const float *__restrict__ pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
  float4 *__restrict__ rec = sorted + static_cast<size_t>(blockIdx.x) * n;  // one record array per scene
  int32_t *__restrict__ out = idx + static_cast<size_t>(blockIdx.x) * m;

  unsigned int nvalid = 0u;
  bool partner_lost = false;  // NWG = 2: this wave gave up waiting for the other workgroup once; wave-uniform
  // NWG = 2: ONE Morton order per scene.  Workgroup 0 sorts; workgroup 1 waits for it and reads the same records.
  // (Until round 5 both sorted redundantly into copies of their own.  The scatter takes its positions from LDS
  // atomics, so the order of the points INSIDE a cell is whatever order the atomics happened in -- and the two
  // workgroups split the record array by POSITION: a point of the cell the split runs through could be in the first
  // half of one order and in the second half of the other, i.e. owned by both, and another point of that cell by
  // NEITHER, which then never became a sample.  The two orders agree almost always (same code, same timing on two
  // idle CUs), so the indices were wrong about once in a hundred launches, more often next to other kernels: the
  // unexplained whole-step failure of round 4.  tests/test_ops_gpu.py::test_fps_two_workgroups_under_cotenancy_stress.)
  if (NWG == 1 || half == 0) {
  // ---- prologue 1: bounding box of the participating points
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += TH) {
    const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
    if (!fps_skipped<DM>(x, y, z)) {
      lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
      hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
  }
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_reduce_min(lo[a]);
    hi[a] = wave_reduce_max(hi[a]);
  }
  if (lane == 0)
    for (int a = 0; a < 3; ++a) {
      s_red[a][w] = lo[a];
      s_red[3 + a][w] = hi[a];
    }
  for (int c = tid; c < kMortonCells; c += TH) s_hist[c] = 0u;
  if (tid < 3) s_slot[tid] = 0ull;
  __syncthreads();
  float inv[3];
  for (int a = 0; a < 3; ++a) {
    float l = s_red[a][0], h = s_red[3 + a][0];
    for (int q = 1; q < W; ++q) {
      l = fminf(l, s_red[a][q]);
      h = fmaxf(h, s_red[3 + a][q]);
    }
    lo[a] = l;
    inv[a] = (h > l) ? 16.0f / (h - l) : 0.0f;
  }
  auto cell_of = [&](float x, float y, float z) -> uint32_t {
    const int qx = min(15, max(0, static_cast<int>((x - lo[0]) * inv[0])));
    const int qy = min(15, max(0, static_cast<int>((y - lo[1]) * inv[1])));
    const int qz = min(15, max(0, static_cast<int>((z - lo[2]) * inv[2])));
    return spread4(qx) | (spread4(qy) << 1) | (spread4(qz) << 2);
  };

  // ---- prologue 2: counting sort by Morton cell into `rec` (x, y, z, index)
  for (int k = tid; k < n; k += TH) {
    const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
    if (!fps_skipped<DM>(x, y, z)) atomicAdd(&s_hist[cell_of(x, y, z)], 1u);
  }
  __syncthreads();
  {  // exclusive scan of the 4096 counters: PER per thread
    unsigned int v[PER], sum = 0u;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      v[q] = s_hist[tid * PER + q];
      sum += v[q];
    }
    unsigned int incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const unsigned int up = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += up;
    }
    if (lane == kWave - 1) s_wsum[w] = incl;
    __syncthreads();
    unsigned int base = 0u;
    for (int q = 0; q < w; ++q) base += s_wsum[q];
    unsigned int run = base + incl - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      s_hist[tid * PER + q] = run;
      run += v[q];
    }
  }
  __syncthreads();
  for (int q = 0; q < W; ++q) nvalid += s_wsum[q];
  for (int k = tid; k < n; k += TH) {
    const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
    if (!fps_skipped<DM>(x, y, z)) {
      const unsigned int pos = atomicAdd(&s_hist[cell_of(x, y, z)], 1u);
      rec[pos] = make_float4(x, y, z, __uint_as_float(static_cast<uint32_t>(k)));
    }
  }
  if constexpr (NWG == 2) {
    // hand the records over: every thread makes its record stores visible at agent scope (a write-back of this XCD's
    // L2 -- once per launch, not per round), then ONE flag word carries the record count to the partner
    __threadfence();
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(&mail[static_cast<size_t>(gridDim.x) * 4 + blockIdx.x].kk,
                         (static_cast<unsigned long long>(nvalid) << 1) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __threadfence_block();
    __syncthreads();
  }
  } else {
    // workgroup 1 of the pair: wait for the sorted records (the same bounded wait as the rounds' below)
    if (tid < 3) s_slot[tid] = 0ull;
    unsigned long long word = 0ull;
    int gave_up = 0;
    if (lane == 0) {
      int spins = 0;
      do {
        word = __hip_atomic_load(&mail[static_cast<size_t>(gridDim.x) * 4 + blockIdx.x].kk, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      } while (word == 0ull && ++spins < spin_limit);
      if (word == 0ull) {
        gave_up = 1;
        __hip_atomic_store(lost_events, 0x80000000u | ((blockIdx.x & 0x7fffu) << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    word = __shfl(word, 0, kWave);
    partner_lost = __shfl(gave_up, 0, kWave) != 0;
    nvalid = static_cast<unsigned int>(word >> 1);  // (0 when the partner was lost: this workgroup owns nothing)
    __threadfence();  // acquire: nothing of an earlier launch's records may be served from this XCD's caches
    __syncthreads();
  }

  // ---- prologue 3: buckets into registers; bucket b -> wave b % W, slot b / W
  Slots<SL> px, py, pz, t;
  BucketMeta md;
  md.lox = md.loy = md.loz = INFINITY;
  md.hix = md.hiy = md.hiz = -INFINITY;
  md.maxt = -1.0f;
  // NWG = 2: workgroup `half` owns buckets [half * share, (half + 1) * share) of the Morton order
  const unsigned int nbuckets = (nvalid + kWave - 1) / kWave;
  const unsigned int share = NWG == 1 ? nbuckets : (nbuckets + 1) / 2;
  auto load_slot = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const unsigned int bucket = static_cast<unsigned int>(j) * W + w;
    const unsigned int pos = (static_cast<unsigned int>(half) * share + bucket) * kWave + lane;
    float x = 0.f, y = 0.f, z = 0.f, tt = -1.0f;
    uint32_t kk = 0xffffffffu;
    if (pos < nvalid && bucket < share) {
      const float4 r = rec[pos];
      x = r.x; y = r.y; z = r.z;
      const uint32_t k = __float_as_uint(r.w);
      const uint32_t kmod = k & ((1u << log2T) - 1u);
      kk = (bitrev_low(kmod, log2T) << KB) | k;  // order: bitrev(k mod T), then k  (k < 2^KB)
      tt = 1e10f;  // sampling.cpp:75-77
    }
    px.template put<j>(x); py.template put<j>(y); pz.template put<j>(z); t.template put<j>(tt);
    s_keys[j * TH + tid] = kk;
    const bool has = tt >= 0.0f;
    const float bl0 = wave_reduce_min(has ? x : INFINITY), bl1 = wave_reduce_min(has ? y : INFINITY),
                bl2 = wave_reduce_min(has ? z : INFINITY);
    const float bh0 = wave_reduce_max(has ? x : -INFINITY), bh1 = wave_reduce_max(has ? y : -INFINITY),
                bh2 = wave_reduce_max(has ? z : -INFINITY);
    const float bt = wave_reduce_max(tt);
    if (lane == j) {
      md.lox = bl0; md.loy = bl1; md.loz = bl2;
      md.hix = bh0; md.hiy = bh1; md.hiz = bh2;
      md.maxt = bt;
    }
  };
  static_for<SL>(load_slot);

  out_put(s_out, 0, 0);  // :88-89
  if (half == 0) out_flush(s_out, 0, m, out);
  float cx = pts[0], cy = pts[1], cz = pts[2];
  float wv = -1.0f;           // this wave's candidate: max running distance ...
  uint32_t wk = 0xffffffffu;  // ... the key of the point holding it ...
  float wx = 0.f, wy = 0.f, wz = 0.f;  // ... and its coordinates
  int r3 = 1;                 // j % 3
  int wslot = -1;             // the slot of the wave's candidate (-1: none yet)

  FPS_PROF_DECL;
  for (int j = 1; j < m; ++j) {
    // lane s: can bucket s change?  LB = |clamp(c, box) - c|^2 with the rounding of sqdist3
    // (v_med3_f32 = the clamp in one instruction; an empty slot's box is (+inf, -inf): the median is c itself, LB = 0,
    // and 0 < maxt = -1 is false)
    const float qx = __builtin_amdgcn_fmed3f(cx, md.lox, md.hix), qy = __builtin_amdgcn_fmed3f(cy, md.loy, md.hiy),
                qz = __builtin_amdgcn_fmed3f(cz, md.loz, md.hiz);
    const float lb = sqdist3<DM>(__fsub_rn(qx, cx), __fsub_rn(qy, cy), __fsub_rn(qz, cz));
    unsigned long long mask = __ballot(lane < SL && lb < md.maxt);
    const unsigned long long touched = mask;
    FPS_PROF_MARK(0);
    FPS_PROF_COUNT(6, __builtin_popcountll(mask));
    FPS_PROF_COUNT(7, mask != 0ull);
    if (mask != 0ull) {
      do {
        const int sl = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        float wmax;
        if (Slots<SL>::B == 0 || sl < Slots<SL>::A) wmax = bucket_touch<DM>(px.a, py.a, pz.a, t.a, sl, cx, cy, cz);
        else wmax = bucket_touch<DM>(px.b, py.b, pz.b, t.b, sl - Slots<SL>::A, cx, cy, cz);
        if (lane == sl) md.maxt = wmax;
      } while (mask != 0ull);
      FPS_PROF_MARK(1);
    }
    // Running distances only decrease, so the wave's candidate stands unless its own bucket was touched (it always is
    // when the candidate won the previous round: that bucket contains the new sample).
    if (touched != 0ull && (wslot < 0 || ((touched >> wslot) & 1ull) != 0ull)) {
      // the wave's candidate: the largest bucket maximum (lanes >= SL hold -1), and among the points holding it --
      // possibly in several buckets -- the one with the smallest key
      wv = wave_max_f32(md.maxt);
      wk = 0xffffffffu;
      wslot = -1;
      if (wv >= 0.0f) {
        unsigned long long tie = __ballot(md.maxt == wv);
        do {
          const int sl = __builtin_ctzll(tie);
          tie &= tie - 1ull;
          const uint32_t key = s_keys[sl * TH + tid];
          const uint32_t before = wk;
          if (Slots<SL>::B == 0 || sl < Slots<SL>::A) bucket_resolve(px.a, py.a, pz.a, t.a, sl, key, wv, wk, wx, wy, wz);
          else bucket_resolve(px.b, py.b, pz.b, t.b, sl - Slots<SL>::A, key, wv, wk, wx, wy, wz);
          if (wk != before) wslot = sl;
        } while (tie != 0ull);
      }
      FPS_PROF_MARK(2);
    }
    // ---- exchange: lane 0 of every wave posts its coordinates and raises the round's 64-bit maximum
    // (distance, ~key, sender) with one LDS atomic; behind the barrier every wave reads the maximum and -- in the same
    // LDS round trip, lane l fetching wave (l mod W)'s coordinates -- picks the sender's by readlane.  (A barrier-free
    // variant, records with round tags polled by every wave, measured slower: 2.56 vs 2.32 ms -- the polling waves take
    // issue slots from the wave everybody is waiting for.)
    if (lane == 0 && wv >= 0.0f) {
      // the sender id in the low bits never decides (keys are unique) and tells the readers whose coordinates to take
      const unsigned long long kk = (static_cast<unsigned long long>(__float_as_uint(wv) + 1u) << 32) |
                                    ((static_cast<uint32_t>(~wk) & ((1u << (KB + 9)) - 1u)) << IDB) |
                                    static_cast<uint32_t>(w + W * half);
      s_xyz[j & 1][w] = make_float4(wx, wy, wz, 0.f);
      atomicMax(&s_slot[r3], kk);
    }
    FPS_PROF_MARK(3);
    __syncthreads();
    FPS_PROF_MARK(4);
    unsigned long long kk = s_slot[r3];
    const float4 cl = s_xyz[j & 1][lane & (W - 1)];
    const int r3n = r3 == 2 ? 0 : r3 + 1;
    if (tid == 0) s_slot[r3n == 2 ? 0 : r3n + 1] = 0ull;  // (j + 2) % 3: next use is two barriers away
    r3 = r3n;
    // (the builtin returns int: without the casts the low word would be sign-extended over the high one)
    kk = (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(kk >> 32)))) << 32) |
         static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(kk)));
    const int wwin = static_cast<int>(static_cast<uint32_t>(kk) & (W - 1));
    float4 c = make_float4(readlane_f(cl.x, wwin), readlane_f(cl.y, wwin), readlane_f(cl.z, wwin), 0.f);
    if constexpr (NWG == 2) {
      // One relaxed 64-bit atomic per workgroup and round: [distance + 1 : 32][~key : 25][round mod 128 : 7].  No
      // fences (an agent-scope release / acquire pair costs an L2 write-back + invalidate per round): the word is
      // self-contained, and the winner's coordinates are re-read from the (read-only) input by its index.
      constexpr unsigned long long kSeqMask = (1ull << IDB) - 1ull;
      FpsMailbox *mine = mail + (static_cast<size_t>(blockIdx.x) * 2 + (j & 1)) * 2 + half;
      FpsMailbox *theirs = mail + (static_cast<size_t>(blockIdx.x) * 2 + (j & 1)) * 2 + (half ^ 1);
      const unsigned long long seq = static_cast<unsigned long long>(j) & kSeqMask;
      if (tid == 0) __hip_atomic_store(&mine->kk, (kk & ~kSeqMask) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // every wave polls for itself (lane 0) and broadcasts: no second barrier in the round
      unsigned long long pkk = 0ull;
      if (!partner_lost) {
        int gave_up = 0;
        if (lane == 0) {
          int spins = 0;
          do {
            pkk = __hip_atomic_load(&theirs->kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((pkk & kSeqMask) != seq && ++spins < spin_limit);
          if ((pkk & kSeqMask) != seq) {
            // the partner did not answer within the limit: from here on the indices are wrong.  Say so where the host
            // sees it, and never wait again (a wait per round would keep the device busy for m x the limit)
            gave_up = 1;
            pkk = 0ull;
            // (a plain system-scope store, not a read-modify-write: no PCIe atomics needed; the last writer's
            // diagnostic stays: bit 31 | scene << 16 | round)
            __hip_atomic_store(lost_events, 0x80000000u | ((blockIdx.x & 0x7fffu) << 16) | (static_cast<unsigned>(j) & 0xffffu),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        pkk = __shfl(pkk, 0, kWave);
        partner_lost = __shfl(gave_up, 0, kWave) != 0;
      }
      if ((pkk & ~kSeqMask) > (kk & ~kSeqMask)) {  // the partner's candidate wins (larger distance, then smaller key)
        kk = pkk;
        const uint32_t k = (~(static_cast<uint32_t>(pkk) >> IDB)) & ((1u << KB) - 1u);
        c = make_float4(pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2], 0.f);
      }
    }
    if ((kk >> 32) == 0ull) {  // nothing participated: reference besti = 0
      out_put(s_out, j, 0);
      cx = pts[0]; cy = pts[1]; cz = pts[2];
    } else {
      const uint32_t lo32 = static_cast<uint32_t>(kk);
      out_put(s_out, j, static_cast<int32_t>((~(lo32 >> IDB)) & ((1u << KB) - 1u)));
      cx = c.x; cy = c.y; cz = c.z;
    }
    if (half == 0) out_flush(s_out, j, m, out);
#ifdef CODA_FPS_PROF
    asm volatile("" ::"v"(cx), "v"(cy), "v"(cz));
#endif
    FPS_PROF_MARK(5);
  }
  FPS_PROF_STORE;
}

constexpr int kBucketMinPoints = 4096, kBucketMaxPoints = 64 * kBucketWaves * 40, kBucketMinSamples = 128;

bool bucket_eligible(int n, int m) { return n >= kBucketMinPoints && n <= kBucketMaxPoints && m >= kBucketMinSamples; }

template <int SL, int W = kBucketWaves>
int launch_bucket(const float *xyz, int b, int n, int m, int log2T, float4 *ws, int32_t *idx, hipStream_t s) {
  constexpr size_t lds = sizeof(uint32_t) * SL * 64 * W;
  int st = CODA_OK;
  CODA_DISPATCH_DM(distance_mode(), {
    auto kern = fps_bucket_kernel<SL, DM, 1, W>;
    st = raise_dynamic_lds(kern, lds, 20 * 1024);  // static + dynamic LDS exceeds the 64 KB default
    if (st == CODA_OK)
      hipLaunchKernelGGL(kern, dim3(b), dim3(64 * W), lds, s, xyz, n, m, log2T, ws, idx,
                         static_cast<FpsMailbox *>(nullptr), static_cast<unsigned int *>(nullptr), 0, -1);
  });
  return st;
}

// Waves per workgroup of the bucketed kernels: 16 (4 per SIMD, 128 registers per lane, at most 20 slots) or 8 (40
// slots).  More waves spread the buckets a round touches -- a dozen, spatially clustered -- over more issue ports:
// 2.02 vs 2.16 ms for 8 x 20 000 -> 2048 with one workgroup per scene (default 16); with two workgroups per scene the
// round is dominated by the mailbox round trip and 8 waves are slightly ahead (3.58 vs 3.64 ms for 8 x 40 000: default
// 8).  The `waves` argument of coda_furthest_point_sampling_opt_f32 / CODA_FPS_WAVES force one shape for both (A/B, tests).
int default_fps_waves() {
  static const int v = [] {
    const char *e = getenv("CODA_FPS_WAVES");
    return !e ? 0 : (atoi(e) == 8 ? 8 : (atoi(e) == 16 ? 16 : 0));
  }();
  return v;
}
int bucket_waves(int workgroups_per_scene = 1) {
  int v = call_options().fps_waves;  // this call's option (coda_furthest_point_sampling_opt_f32), else the default
  if (v != 8 && v != 16) v = default_fps_waves();
  return v != 0 ? v : (workgroups_per_scene == 2 ? 8 : 16);
}

int dispatch_bucket(const float *xyz, int b, int n, int m, int log2T, float4 *ws, int32_t *idx, hipStream_t s) {
  if (bucket_waves() == 16) {
    const int sl = ceil_div(n, 64 * 16);
    if (sl <= 8) return launch_bucket<8, 16>(xyz, b, n, m, log2T, ws, idx, s);
    if (sl <= 12) return launch_bucket<12, 16>(xyz, b, n, m, log2T, ws, idx, s);
    if (sl <= 16) return launch_bucket<16, 16>(xyz, b, n, m, log2T, ws, idx, s);
    return launch_bucket<20, 16>(xyz, b, n, m, log2T, ws, idx, s);
  }
  const int sl = ceil_div(n, 64 * kBucketWaves);
  if (sl <= 8) return launch_bucket<8>(xyz, b, n, m, log2T, ws, idx, s);
  if (sl <= 16) return launch_bucket<16>(xyz, b, n, m, log2T, ws, idx, s);
  if (sl <= 24) return launch_bucket<24>(xyz, b, n, m, log2T, ws, idx, s);
  if (sl <= 32) return launch_bucket<32>(xyz, b, n, m, log2T, ws, idx, s);
  return launch_bucket<40>(xyz, b, n, m, log2T, ws, idx, s);
}

// two workgroups per scene: 20 480 < n <= 40 960
// CODA_FPS_COOP_MIN=<n>: use the pair from n points on (A/B; default: only where one workgroup cannot hold the cloud)
int bucket2_min_points() {
  static const int v = [] { const char *e = getenv("CODA_FPS_COOP_MIN"); return e ? atoi(e) : kBucketMaxPoints + 1; }();
  return v;
}
bool bucket2_eligible(int n, int m) {
  return n >= bucket2_min_points() && n >= 2 * kBucketMinPoints && n <= 2 * kBucketMaxPoints && m >= kBucketMinSamples;
}
// one Morton-ordered record array per scene (round 5 on: workgroup 0 sorts, workgroup 1 reads the same records), then the mailboxes
size_t bucket2_mail_offset(int b, int n) { return (sizeof(float4) * static_cast<size_t>(b) * n + 255) & ~static_cast<size_t>(255); }
// per scene: 2 round parities x 2 workgroups of candidate mailboxes, then (after all of those) one hand-over word
constexpr int kMailboxesPerScene = 5;
constexpr int kPairScenesPerLaunch = 64;  // see launch_bucket2
size_t bucket2_workspace_bytes(int b, int n) {
  return bucket2_mail_offset(b, n) + sizeof(FpsMailbox) * kMailboxesPerScene * static_cast<size_t>(b);
}

template <int SL, int W = kBucketWaves>
int launch_bucket2(const float *xyz, int b, int n, int m, int log2T, void *ws, int32_t *idx, hipStream_t s) {
  constexpr size_t lds = sizeof(uint32_t) * SL * 64 * W;
  FpsMailbox *mail = reinterpret_cast<FpsMailbox *>(static_cast<char *>(ws) + bucket2_mail_offset(b, n));
  const FpsStatusWord &status = fps_status_word(true);
  if (!status.host) return status.err ? status.err : CODA_ENOSPC;  // no way to report a lost partner: do not launch
  const CallOptions &o = call_options();
  const int spin_limit = o.fps_spin_limit > 0 ? o.fps_spin_limit : kFpsSpinLimit;
  hipError_t e = hipMemsetAsync(mail, 0, sizeof(FpsMailbox) * kMailboxesPerScene * static_cast<size_t>(b), s);  // round numbers start at 1
  if (e != hipSuccess) return static_cast<int>(e);
  int st = CODA_OK;
  // LAUNCH-SHAPE PRECONDITION of the pair: workgroups (s, 0) and (s, 1) wait for each other, so both must be resident
  // at the same time.  The grid is (scenes, 2) with x fastest -- the (s, 0) of ALL scenes are dispatched first, and
  // with more scenes than the chip has workgroup slots (one per CU at this LDS / register footprint: 256) every
  // resident workgroup would be an (s, 0) spinning for an (s, 1) that cannot start (loud since
  // round 5 -- CODA_ELOST -- but a precondition nothing checked).  So: at most kPairScenesPerLaunch scenes per
  // launch (2 x 64 workgroups = half the chip; a multiple of 8, so that (s, 0) and (s, 1) = linear ids s and
  // s + scenes still share an XCD), more scenes = more launches on the same stream.
  CODA_DISPATCH_DM(distance_mode(), {
    auto kern = fps_bucket_kernel<SL, DM, 2, W>;
    st = raise_dynamic_lds(kern, lds, 20 * 1024);
    for (int s0 = 0; st == CODA_OK && s0 < b; s0 += kPairScenesPerLaunch) {
      const int nb = b - s0 < kPairScenesPerLaunch ? b - s0 : kPairScenesPerLaunch;
      hipLaunchKernelGGL(kern, dim3(nb, 2), dim3(64 * W), lds, s, xyz + static_cast<size_t>(s0) * n * 3, n, m, log2T,
                         static_cast<float4 *>(ws) + static_cast<size_t>(s0) * n, idx + static_cast<size_t>(s0) * m,
                         mail + static_cast<size_t>(s0) * kMailboxesPerScene, status.host, spin_limit, o.fps_drop_half);
    }
  });
  return st;
}

int dispatch_bucket2(const float *xyz, int b, int n, int m, int log2T, void *ws, int32_t *idx, hipStream_t s) {
  const int share = ceil_div(ceil_div(n, 64), 2);  // buckets per workgroup
  if (bucket_waves(2) == 16) {
    const int sl = ceil_div(share, 16);
    if (sl <= 8) return launch_bucket2<8, 16>(xyz, b, n, m, log2T, ws, idx, s);
    if (sl <= 12) return launch_bucket2<12, 16>(xyz, b, n, m, log2T, ws, idx, s);
    if (sl <= 16) return launch_bucket2<16, 16>(xyz, b, n, m, log2T, ws, idx, s);
    return launch_bucket2<20, 16>(xyz, b, n, m, log2T, ws, idx, s);
  }
  const int sl = ceil_div(share, kBucketWaves);
  if (sl <= 8) return launch_bucket2<8>(xyz, b, n, m, log2T, ws, idx, s);
  if (sl <= 16) return launch_bucket2<16>(xyz, b, n, m, log2T, ws, idx, s);
  if (sl <= 24) return launch_bucket2<24>(xyz, b, n, m, log2T, ws, idx, s);
  if (sl <= 32) return launch_bucket2<32>(xyz, b, n, m, log2T, ws, idx, s);
  return launch_bucket2<40>(xyz, b, n, m, log2T, ws, idx, s);
}

// ---- streaming fallback: any n; running distances in LDS or in workspace -------
template <int THREADS, int DM>
__global__ __launch_bounds__(THREADS) void fps_stream_kernel(const float *__restrict__ xyz, int n,
                                                             int m, int log2T,
                                                             float *__restrict__ temp_global,
                                                             int32_t *__restrict__ idx) {
  constexpr int NW = THREADS / kWave;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint2(*s_key)[NW] = reinterpret_cast<uint2(*)[NW]>(smem);
  float *temp = temp_global ? temp_global + static_cast<size_t>(blockIdx.x) * n
                            : reinterpret_cast<float *>(smem + sizeof(uint2) * 2 * NW);

  const int tid = threadIdx.x;
  const float *__restrict__ pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
  int32_t *__restrict__ out = idx + static_cast<size_t>(blockIdx.x) * m;

  for (int k = tid; k < n; k += THREADS)
    temp[k] = fps_skipped<DM>(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2]) ? -1.0f : 1e10f;
  if (tid == 0) out[0] = 0;
  float cx = pts[0], cy = pts[1], cz = pts[2];

  for (int j = 1; j < m; ++j) {
    float best = -1.0f;
    int bestk = 0;
    for (int k = tid; k < n; k += THREADS) {  // each thread only touches its own temp[k]
      const float d = sqdist3<DM>(__fsub_rn(pts[k * 3 + 0], cx), __fsub_rn(pts[k * 3 + 1], cy),
                              __fsub_rn(pts[k * 3 + 2], cz));
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      if (d2 > best) {
        best = d2;
        bestk = k;
      }
    }
    uint32_t hi = 0u, lo = 0u;
    if (best >= 0.0f) {
      hi = __float_as_uint(best) + 1u;
      lo = ~tie_rank(static_cast<uint32_t>(bestk), log2T);
    }
    const uint32_t old = block_argmax<NW>(hi, lo, s_key, j & 1, log2T);
    if (tid == 0) out[j] = static_cast<int32_t>(old);
    cx = pts[old * 3 + 0];
    cy = pts[old * 3 + 1];
    cz = pts[old * 3 + 2];
  }
}

template <int P, int THREADS>
void launch_reg(const float *xyz, int b, int n, int m, int log2T, int32_t *idx, hipStream_t s) {
  CODA_DISPATCH_DM(distance_mode(), hipLaunchKernelGGL((fps_reg_kernel<P, THREADS, DM>), dim3(b), dim3(THREADS), 0,
                                                        s, xyz, n, m, log2T, idx));
}

// The thread-local strict '>' scan is only rank-ordered when every point of a
// thread shares (k mod T), i.e. when T divides THREADS: 256 threads for
// n < 512 (T <= 256), 512 threads up to 4096 points, 1024 threads (16 waves =
// 4 per SIMD, 128 VGPRs each) up to 24 points per thread.
bool dispatch_reg(const float *xyz, int b, int n, int m, int log2T, int32_t *idx, hipStream_t s) {
  if (n < 512) {
    if (n <= 256) launch_reg<1, 256>(xyz, b, n, m, log2T, idx, s);
    else launch_reg<2, 256>(xyz, b, n, m, log2T, idx, s);
    return true;
  }
  if (n <= 4096) {
    const int p = ceil_div(n, 512);
    if (p <= 1) launch_reg<1, 512>(xyz, b, n, m, log2T, idx, s);
    else if (p <= 2) launch_reg<2, 512>(xyz, b, n, m, log2T, idx, s);
    else if (p <= 4) launch_reg<4, 512>(xyz, b, n, m, log2T, idx, s);
    else launch_reg<8, 512>(xyz, b, n, m, log2T, idx, s);
    return true;
  }
  const int p = ceil_div(n, 1024);
  if (p <= 8) launch_reg<8, 1024>(xyz, b, n, m, log2T, idx, s);
  else if (p <= 12) launch_reg<12, 1024>(xyz, b, n, m, log2T, idx, s);
  else if (p <= 16) launch_reg<16, 1024>(xyz, b, n, m, log2T, idx, s);
  else if (p <= 20) launch_reg<20, 1024>(xyz, b, n, m, log2T, idx, s);
  else if (p <= 24) launch_reg<24, 1024>(xyz, b, n, m, log2T, idx, s);
  else return false;
  return true;
}

// include/cuda_utils.h:17-21, evaluated exactly like the reference (double log).
int reference_block_log2(int n) {
  int pow_2 = static_cast<int>(std::log(static_cast<double>(n)) / std::log(2.0));
  if (pow_2 > 9) pow_2 = 9;  // TOTAL_THREADS = 512
  if (pow_2 < 0) pow_2 = 0;
  return pow_2;
}

constexpr int kStreamThreads = 1024;
constexpr size_t kLdsBudget = 160 * 1024;
constexpr size_t kStreamKeyBytes = sizeof(uint2) * 2 * (kStreamThreads / kWave);

}  // namespace
}  // namespace coda

CODA_API size_t coda_furthest_point_sampling_workspace_bytes(int b, int n, int m) {
  if (b <= 0 || n <= 0) return 0;
  if (coda::bucket2_eligible(n, m)) return coda::bucket2_workspace_bytes(b, n);  // one record array per scene + the mailboxes
  if (coda::bucket_eligible(n, m)) return sizeof(float4) * static_cast<size_t>(b) * n;  // Morton-sorted records
  const size_t lds_need = coda::kStreamKeyBytes + sizeof(float) * static_cast<size_t>(n);
  if (n <= 1024 * 24 || lds_need <= coda::kLdsBudget) return 0;
  return sizeof(float) * static_cast<size_t>(b) * n;
}

CODA_API int coda_furthest_point_sampling_f32(const float *xyz, int b, int n, int m, int32_t *idx,
                                              void *workspace, size_t workspace_bytes,
                                              void *stream) {
  using namespace coda;
  if (b < 0 || n <= 0 || m < 0 || (b > 0 && (!xyz || (m > 0 && !idx)))) return CODA_EINVAL;
  if (b == 0 || m == 0) return CODA_OK;  // sampling_gpu.cu:75
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int log2T = reference_block_log2(n);
  // an EARLIER two-workgroup launch of this process gave up waiting for its partner workgroup: its indices are wrong.
  // Sticky until read with coda_fps_lost_partner_events(1) -- nothing is launched on top of an unacknowledged loss
  if (fps_lost_events(false) != 0u) return CODA_ELOST;

  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated HIP calls
  // variant 0 (default): bucketed kernel (needs the workspace), else v2 where it applies;
  // 1: force the v1 register kernel, 2: never use the bucketed kernel (A/B, tests)
  static const int variant = [] { const char *e = getenv("CODA_FPS_VARIANT"); return e ? atoi(e) : 0; }();
  bool done = false;
  if (variant == 0 && bucket2_eligible(n, m) && workspace && workspace_bytes >= bucket2_workspace_bytes(b, n) &&
      (reinterpret_cast<uintptr_t>(workspace) & 255) == 0) {
    const int st = dispatch_bucket2(xyz, b, n, m, log2T, workspace, idx, s);
    if (st != CODA_OK) return st;
    done = true;
  }
  if (!done && variant == 0 && bucket_eligible(n, m) && workspace &&
      workspace_bytes >= sizeof(float4) * static_cast<size_t>(b) * n && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
    const int st = dispatch_bucket(xyz, b, n, m, log2T, static_cast<float4 *>(workspace), idx, s);
    if (st != CODA_OK) return st;
    done = true;
  }
  if (!done) done = variant != 1 && dispatch_t512(xyz, b, n, m, idx, s);
  if (!done) done = dispatch_reg(xyz, b, n, m, log2T, idx, s);
  if (!done) {
    const size_t lds_need = kStreamKeyBytes + sizeof(float) * static_cast<size_t>(n);
    const bool in_lds = lds_need <= kLdsBudget;
    if (!in_lds && (!workspace || workspace_bytes < sizeof(float) * static_cast<size_t>(b) * n)) return CODA_ENOSPC;
    int st = CODA_OK;
    CODA_DISPATCH_DM(distance_mode(), {
      auto kern = fps_stream_kernel<kStreamThreads, DM>;
      if (in_lds) {
        st = raise_dynamic_lds(kern, lds_need);
        if (st == CODA_OK)
          hipLaunchKernelGGL(kern, dim3(b), dim3(kStreamThreads), lds_need, s, xyz, n, m, log2T,
                             static_cast<float *>(nullptr), idx);
      } else {
        hipLaunchKernelGGL(kern, dim3(b), dim3(kStreamThreads), kStreamKeyBytes, s, xyz, n, m, log2T,
                           static_cast<float *>(workspace), idx);
      }
    });
    if (st != CODA_OK) return st;
  }
  return launch_status();
}

CODA_API int coda_furthest_point_sampling_opt_f32(const float *xyz, int b, int n, int m, int32_t *idx, void *workspace,
                                                  size_t workspace_bytes, int distance_mode, int waves, void *stream) {
  if (distance_mode < -1 || distance_mode >= coda::kDistanceModes) return CODA_EINVAL;
  if (waves != 0 && waves != 8 && waves != 16) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.distance_mode = distance_mode;
  o.fps_waves = waves;
  coda::ScopedCallOptions scope(o);
  return coda_furthest_point_sampling_f32(xyz, b, n, m, idx, workspace, workspace_bytes, stream);
}

CODA_API unsigned int coda_fps_lost_partner_events(int reset) { return coda::fps_lost_events(reset != 0); }

CODA_API int coda_furthest_point_sampling_dbg_f32(const float *xyz, int b, int n, int m, int32_t *idx, void *workspace,
                                                  size_t workspace_bytes, int spin_limit, int drop_half, void *stream) {
  if (spin_limit < 0 || drop_half < -1 || drop_half > 1) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.fps_spin_limit = spin_limit;
  o.fps_drop_half = drop_half;
  coda::ScopedCallOptions scope(o);
  return coda_furthest_point_sampling_f32(xyz, b, n, m, idx, workspace, workspace_bytes, stream);
}

#ifdef CODA_FPS_PROF
CODA_API int coda_fps_prof_read(unsigned long long *host) {
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(coda::g_fps_prof), sizeof(coda::g_fps_prof)));
}
#endif
