// attention_common.hip.h -- parameter blocks and small device helpers shared by the fp32-MFMA
// (attention.hip) and bf16-MFMA (attention_bf16.hip) attention kernels.
#pragma once
#include <hip/hip_ext.h>

#include <cstdlib>

#include "common.hip.h"

namespace coda {

// Measurement aid (coda_mha_timing_*): while a KernelTimer is alive on this thread, the NEXT kernel launched through
// mha_launch() carries the timer's event pair as its own start / stop events (hipExtLaunchKernelGGL: the timestamps of
// the dispatch itself, what rocprofv3's kernel trace reports).  Events recorded around the launch instead measured the
// launch gap and two marker packets too: +3 us on a 10-20 us decoder kernel (round 3's "event-timed vs traced").
struct PendingTiming {
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
PendingTiming &pending_timing();  // thread-local, defined in attention.hip

template <class K, class... Args>
inline void mha_launch(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
  PendingTiming &t = pending_timing();
  if (t.e0) {
    hipExtLaunchKernelGGL(kern, grid, block, static_cast<uint32_t>(lds), stream, t.e0, t.e1, 0, args...);
    t.e0 = t.e1 = nullptr;
  } else {
    hipLaunchKernelGGL(kern, grid, block, lds, stream, args...);
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kTile = 32;  // keys (or queries) per MFMA tile
constexpr float kLog2e = 1.4426950408889634f;
// forward kernels: how far (log2 units) a row maximum may outgrow the reference point of its exponentials before the
// accumulators are rescaled (probabilities stay <= 2^8; see mha_fwd_kernel)
constexpr float kMaxSlack = 8.0f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Counter-based dropout, identical in forward and backward: one 32-bit hash of (seed, b*h, query, key >> 1)
// serves the two keys of an aligned pair, 16 bits each, so the kernels whose lanes hold a query and whose
// registers hold consecutive keys (forward, dQ) pay one hash per two probabilities.
// The mixer is two rounds of (24-bit multiply-add, xor-shift): eight FULL-RATE integer instructions.  fp32 MFMAs
// and VALU instructions do not overlap on this part (tools/mfma_valu_probe.hip: a wave pair takes the SUM of its
// MFMA and VALU times), so every VALU slot of the soft-max is paid in full; the previous lowbias32 mixer spent 14
// slots, 8 of them in two quarter-rate v_mul_lo_u32.  Keep rate, adjacent-key / adjacent-query / cross-head /
// cross-seed correlations and row / column rate spreads are indistinguishable from lowbias32 on 1024 x 2048 grids
// (development check, tests/test_attention_gpu.py checks rates and determinism on the device).
__device__ __forceinline__ uint32_t drop_hash(uint32_t c, uint32_t q, uint32_t s_len, uint32_t key) {
  uint32_t x = (q * s_len + (key & ~1u)) ^ c;
  x = __umul24(x, 0xD2B74Fu) + (x >> 8);
  x ^= x >> 15;
  x = __umul24(x, 0xC2B2AFu) + (x >> 11);
  x ^= x >> 14;
  return x;
}
__device__ __forceinline__ uint32_t drop_const(uint32_t seed, uint32_t bh) { return (bh * 0x9E3779B9u) ^ seed ^ (bh << 27); }
__device__ __forceinline__ bool drop_keep_lo(uint32_t h, uint32_t thresh16) { return (h & 0xffffu) >= thresh16; }
__device__ __forceinline__ bool drop_keep_hi(uint32_t h, uint32_t thresh16) { return (h >> 16) >= thresh16; }
__device__ __forceinline__ bool drop_keep(uint32_t h, uint32_t key, uint32_t thresh16) {
  return ((key & 1u) ? (h >> 16) : (h & 0xffffu)) >= thresh16;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct MhaParams {
  const float *q, *k, *v;
  const uint8_t *mask;
  float *out, *lse;
  int b, h, l, s;
  int ldq, ldk, ldv;  // floats between consecutive batch rows of q / k / v (H*D when dense)
  float scale, inv_keep;
  uint32_t thresh16, seed;
  const uint64_t *seed_dev;  // optional device-resident seed (graph replays draw fresh masks)
  int xcd_map;               // XCD-aware workgroup -> (tile, head) mapping (tile_head())
};

// XCD-aware workgroup -> (tile, batch*head) mapping.  Workgroups are dispatched round-robin over the 8
// XCDs in linear-id order (x fastest), so with the plain grid (x = tile, y = head) the tiles of one head
// are spread over all 8 XCDs and every private L2 pulls every head's K/V (or Q/dO) from the fabric:
// FETCH_SIZE showed 2.8-5x the algorithmic bytes.  Here the workgroups that share a head share an XCD.
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding global stores
// (s_waitcnt vmcnt(0)), which a kernel that streams results out between barriers must not do.
__device__ __forceinline__ void lds_only_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct TileHead {
  int tile, bh;
};
__device__ __forceinline__ TileHead tile_head(int xcd_map) {
  const int T = gridDim.x, BH = gridDim.y;
  if (!xcd_map || (BH & 7) != 0) return {static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y)};
  const int i = blockIdx.x + T * blockIdx.y;
  const int j = i >> 3;
  return {j % T, (j / T) * 8 + (i & 7)};
}

__device__ __forceinline__ uint32_t effective_seed(uint32_t seed, const uint64_t *seed_dev) {
  if (!seed_dev) return seed;
  const uint64_t v = *seed_dev;
  return seed ^ static_cast<uint32_t>(v) ^ static_cast<uint32_t>(v >> 32) * 0x9E3779B9u;
}

// ---------------------------------------------------------------------------------------
// Backward.  delta[b,h,q] = sum_dv dout * out  (row-wise), then
//   P  = exp(scale*QK^T - lse)         Pd = dropout(P)
//   dV = Pd^T dO      dP = dO V^T (dropout-masked, /keep)     dS = P * (dP - delta) * scale
//   dK = dS^T Q       dQ = dS K
struct MhaBwdParams {
  const float *q, *k, *v, *out, *lse, *dout;
  const uint8_t *mask;
  float *dq, *dk, *dv, *delta;
  int b, h, l, s;
  int ldq, ldk, ldv;
  int lddq, lddk, lddv;  // floats between consecutive batch rows of dq / dk / dv (H*D when dense)
  int xcd_map;
  float scale, inv_keep;
  uint32_t thresh16, seed;
  const uint64_t *seed_dev;
  int parts = 7;  // which launches a backward call issues: 1 = delta, 2 = dK/dV, 4 = dQ (host-side only)
  int fuse_delta = 0;  // 1: the dQ kernel forms delta = rowsum(dO * O) itself and writes it for the dK/dV kernel behind it
  float *ds = nullptr;  // optional (B*H, L, S) workspace: the dK/dV kernel leaves dS there and dQ = dS K becomes one GEMM
};


// Two query tiles per workgroup in the LDS-staged split-key kernels for long key sequences: 1 when that many
// workgroups (`pairs`) still fill the chip, 2 when they do with the keys in two halves on top (dQ only), else 0.
// CODA_ATTN_QT=0 switches it off (A/B).
inline int split_query_tiles(int pairs) {
  static const bool on = [] { const char *e = getenv("CODA_ATTN_QT"); return !e || atoi(e) != 0; }();
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    return n;
  }();
  if (!on) return 0;
  return pairs >= cus ? 1 : (2 * pairs >= cus ? 2 : 0);
}

// attention_bf16.hip: the same three kernels with bf16 MFMA operands (fp32 tensors, fp32 accumulation
// and softmax); launched by attention.hip's dispatchers inside their timing brackets.
int mha_fwd_bf16(const MhaParams &p, int d, hipStream_t s);
// the same kernels with every fp32 operand carried as three bf16 pieces (fp32-level results, MFMA dtype 2)
bool mha_x3_takes_fwd(const MhaParams &p, int d);
bool mha_x3_takes_dkv(const MhaBwdParams &p, int d);
bool mha_x3_takes_dq(const MhaBwdParams &p, int d);
int mha_fwd_x3(const MhaParams &p, int d, hipStream_t s);
int mha_bwd_dkv_x3(const MhaBwdParams &p, int d, hipStream_t s);
int mha_bwd_dq_x3(const MhaBwdParams &p, int d, hipStream_t s);
int mha_bwd_dkv_bf16(const MhaBwdParams &p, int d, hipStream_t s);
int mha_bwd_dq_bf16(const MhaBwdParams &p, int d, hipStream_t s);

}  // namespace coda
