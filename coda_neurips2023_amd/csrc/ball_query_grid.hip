// ball_query_grid.hip -- cell-binned radius search for gfx950.
//
// The brute-force scan (ball_query.hip) performs M*N distance tests per scene
// (41 M at N=20000, M=2048) and is VALU-bound at ~1 % of the HBM roofline.  The
// algorithmic traffic of the operator is only 12N + 12M + 4MS bytes, so the
// work has to shrink, not the bytes: points are binned into a uniform grid of
// cell size >= radius and a centre only tests the 3x3x3 cells around it
// (~10^2 candidates instead of 2*10^4).
//
// The reference's result is order-dependent -- "the first nsample hits in
// ascending point index, padded with the first hit" (ball_query_gpu.cu:30-43)
// -- so the query kernel collects ALL hits of a ball (any order), ranks them by
// point index with an all-pairs count in LDS, and emits ranks < nsample; the
// output is bit-identical to the serial scan.
//
// Grid: the integer lattice floor(p / cell), cell = 1.001 * radius, folded onto a
// 32 x 32 x 16 torus (cell index = lattice coordinate mod 32 / 32 / 16).  No bounding box
// is needed (round 1 spent a third of the build on one) and the grid never has to "grow
// until it fits": lattice cells a period apart share a table entry, their points are then
// candidates of each other's centres and fail the distance test like any other miss
// (6.4 m x 6.4 m x 3.2 m at r = 0.2: an indoor scene does not alias at all).
//
// Build (ONE launch, 8 workgroups per scene, no inter-workgroup communication): see grid_build_kernel.
// Round 1's build was one workgroup per scene reading the cloud three times (bounding box, histogram,
// scatter) and scanning 32768 counters: 44-65 us on 8 of 256 CUs.
// Workspace per scene: cell_start[16385] | pad | records[N] (float4).
// Query, nsample <= 64 (round 5, grid_query8_kernel): a wave serves eight centres, eight lanes each, on a flattened
//   candidate list, and a DPP bitonic sorting network produces "the first nsample in index order"; see the kernel.
// Query, larger nsample (rounds 2-4, grid_query_kernel: one wave per centre): lanes 0..8 fetch the nine x-contiguous
//   cell ranges (18 half-ranges when the three x cells wrap around the torus), a wave prefix sum flattens them into one
//   candidate list so that every lane of every 64-wide chunk has work, hits are compacted with ballot/mbcnt into an LDS
//   buffer, ranked, and written as one 256-B row (+ the centred / normalised xyz of the fused QueryAndGroup path).
// Build at 24+ scenes: one 1024-thread workgroup per scene with the whole table in LDS (grid_build1_kernel).
#include "common.hip.h"

#include <cstdlib>

namespace coda {

constexpr int kGridX = 32, kGridY = 32, kGridZ = 16;   // cells per axis of the torus (powers of two)
constexpr int kCells = kGridX * kGridY * kGridZ;       // 16384 table entries per scene
constexpr int kBuildThreads = 512;    // 8 waves = 2 per SIMD: 256 VGPRs per thread for the resident points
constexpr int kKeepMax = 12;          // groups of four points per thread the build keeps in registers (n <= 24576)
constexpr int kHitCap = 256;          // LDS hit records per wave (4 KiB)
constexpr int kQueryWaves = 4;
constexpr int kGridMinPoints = 1024;  // below this the scan kernel is cheaper than building a grid
constexpr int kGridMaxSample = 128;   // nsample + one 64-wide chunk must fit kHitCap

inline size_t grid_scene_bytes(int n) {
  return sizeof(int) * (kCells + 4) + sizeof(float4) * static_cast<size_t>(n);
}

namespace {

// Lattice coordinate along one axis: floor(v * inv_cell) as ONE saturating conversion
// (v_cvt_flr_i32_f32: floor + convert, +-Inf / overflow clamp to INT_MIN / INT_MAX, NaN gives 0).
// Evaluated identically for points and centres and monotone in v, so a hit
// (|dv| < r(1 + 2^-23) <= cell / (1 + 1e-3), see the distance-mode note in common.hip.h) lies within
// +-1 lattice cell of the centre's cell.  Non-finite points simply land in SOME cell: they are
// candidates there and fail `d2 < r2` like in the reference's scan (every comparison with NaN is false,
// Inf - x = Inf), so they need no special casing.
__device__ __forceinline__ int lattice(float v, float inv_cell) {
  int r;
  const float u = __fmul_rn(v, inv_cell);
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(u));
  return r;
}
// lattice coordinate -> torus coordinate (two's complement & = mathematical mod for negatives too)
__device__ __forceinline__ int fold_x(int c) { return c & (kGridX - 1); }
__device__ __forceinline__ int fold_y(int c) { return c & (kGridY - 1); }
__device__ __forceinline__ int fold_z(int c) { return c & (kGridZ - 1); }
__device__ __forceinline__ int cell_of(float x, float y, float z, float inv_cell) {
  return (fold_z(lattice(z, inv_cell)) * kGridY + fold_y(lattice(y, inv_cell))) * kGridX + fold_x(lattice(x, inv_cell));
}

struct SceneWs {
  int *cell_start;
  float4 *records;
};
__device__ __forceinline__ SceneWs scene_ws(unsigned char *ws, size_t scene_stride, int scene) {
  unsigned char *base = ws + static_cast<size_t>(scene) * scene_stride;
  SceneWs r;
  r.cell_start = reinterpret_cast<int *>(base);
  r.records = reinterpret_cast<float4 *>(base + sizeof(int) * (kCells + 4));
  return r;
}

// -DCODA_BQ_PROF (tools/bq_prof.py builds a private copy with it): shader-clock sums of the kernel's phases over all
// waves, read back with coda_bq_prof_read.  Compiles to nothing in the library.
#ifdef CODA_BQ_PROF
constexpr int kBqProfWaves = 16384;
__device__ unsigned long long g_bq_prof[kBqProfWaves][12];  // per wave: six phase clock sums, then counters
#define BQ_PROF_DECL                                                                  \
  unsigned long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                \
  unsigned long long prof_t_ = __builtin_readcyclecounter()
#define BQ_PROF_MARK(i)                                                               \
  do {                                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();                     \
    prof_[i] += now_ - prof_t_;                                                       \
    prof_t_ = now_;                                                                   \
  } while (0)
#define BQ_PROF_COUNT(i, v) prof_[i] += static_cast<unsigned long long>(v)
#define BQ_PROF_STORE                                                                 \
  do {                                                                                \
    const int wid_ = blockIdx.x * (blockDim.x / kWave) + BQ_PROF_WAVE + BQ_PROF_BASE;                                       \
    if (lane == 0 && wid_ < kBqProfWaves)                                             \
      for (int q_ = 0; q_ < 12; ++q_) g_bq_prof[wid_][q_] = prof_[q_];                \
  } while (0)
#else
#define BQ_PROF_DECL
#define BQ_PROF_MARK(i)
#define BQ_PROF_COUNT(i, v)
#define BQ_PROF_STORE
#endif

// ---- build -------------------------------------------------------------------------------------
// kSlabs workgroups per scene; workgroup j owns the table entries [j, j+1) * kCells / kSlabs.  Every
// workgroup walks ALL points of the scene (from L2 after the first one; 4 points = three 16-B loads per
// thread and step), counts its own cells with LDS atomics and, in a register, the points that fall into
// lower slabs -- that count is its base offset, so the slabs need no communication at all.  Scan of its
// 2048 counters, then the scatter of its own points with returning LDS atomics.  G = groups of four
// points a thread keeps in registers across the two passes (0: the cloud is read again).
constexpr int kSlabs = 8;
constexpr int kSlabCells = kCells / kSlabs;               // 2048
constexpr int kPerThread = kSlabCells / kBuildThreads;    // 4 counters per thread in the scan

struct Quad {  // four consecutive points
  float v[12];
};
__device__ __forceinline__ Quad load_quad(const float *__restrict__ pts, int g, int n, bool vec) {
  Quad q;
  const int k0 = 4 * g;
  if (vec && k0 + 3 < n) {
    const float4 a = *reinterpret_cast<const float4 *>(pts + k0 * 3), b = *reinterpret_cast<const float4 *>(pts + k0 * 3 + 4),
                 c = *reinterpret_cast<const float4 *>(pts + k0 * 3 + 8);
    q.v[0] = a.x; q.v[1] = a.y; q.v[2] = a.z; q.v[3] = a.w; q.v[4] = b.x; q.v[5] = b.y; q.v[6] = b.z; q.v[7] = b.w;
    q.v[8] = c.x; q.v[9] = c.y; q.v[10] = c.z; q.v[11] = c.w;
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) q.v[i] = (k0 * 3 + i < n * 3) ? pts[k0 * 3 + i] : 0.0f;  // past the end: index-checked below
  }
  return q;
}

template <int G>
__global__ __launch_bounds__(kBuildThreads) void grid_build_kernel(const float *__restrict__ xyz, int n, float inv_cell,
                                                                  unsigned char *__restrict__ ws,
                                                                  size_t scene_stride, int vec, int nscenes) {
  __shared__ __attribute__((aligned(16))) int s_cnt[kSlabCells];
  __shared__ int s_red[2][kBuildThreads / kWave];
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  // XCD-aware: workgroup g runs on XCD g % 8 (observed dispatch order), so scene = g % B keeps the eight slab
  // workgroups of a scene on ONE XCD -- the cloud is fetched from HBM once instead of once per slab (PMC: 19.5 MB
  // -> 2 MB per call at B = 8).  Only speed depends on the placement.
  const int scene = blockIdx.x % nscenes, slab = blockIdx.x / nscenes;
  const int lo = slab * kSlabCells;
  const float *__restrict__ pts = xyz + static_cast<size_t>(scene) * n * 3;
  const SceneWs w = scene_ws(ws, scene_stride, scene);
  const int ngroups = (n + 3) / 4;

  BQ_PROF_DECL;
  for (int c = tid; c < kSlabCells; c += kBuildThreads) s_cnt[c] = 0;
  constexpr int GG = G > 0 ? G : 1;
  Quad q[GG];
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) q[i] = load_quad(pts, tid + i * kBuildThreads, n, vec != 0);
  }
  __syncthreads();
  BQ_PROF_MARK(0);
  // ---- count my cells; count the points of lower slabs
  int below = 0;
  // A slab is two z layers of the table (the cell index is z-major), so the z coordinate alone says whether a point is
  // below, in or above this workgroup's slab: three instructions for the 7 of 8 points that are not its own instead of
  // the full cell index (round 5: the two passes over ALL points are what the build's time is made of).
  constexpr int kSlabLayers = kGridZ / kSlabs;
  static_assert(kSlabLayers * kGridY * kGridX == kSlabCells, "slabs are whole z layers");
  const int zlo = slab * kSlabLayers;
  auto count_quad = [&](const Quad &qq, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = qq.v[3 * j], y = qq.v[3 * j + 1], z = qq.v[3 * j + 2];
      if (4 * g + j < n) {
        const int dz = fold_z(lattice(z, inv_cell)) - zlo;
        if (dz < 0) ++below;
        else if (dz < kSlabLayers)
          atomicAdd(&s_cnt[(dz * kGridY + fold_y(lattice(y, inv_cell))) * kGridX + fold_x(lattice(x, inv_cell))], 1);
      }
    }
  };
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) count_quad(q[i], tid + i * kBuildThreads);
    // groups beyond the resident ones (a cloud larger than G * 2048 points) are read again in each pass
    for (int g = tid + GG * kBuildThreads; g < ngroups; g += kBuildThreads) count_quad(load_quad(pts, g, n, vec != 0), g);
  } else {
    for (int g = tid; g < ngroups; g += kBuildThreads) count_quad(load_quad(pts, g, n, vec != 0), g);
  }
  BQ_PROF_MARK(1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off);
  if (lane == 0) s_red[0][wv] = below;
  __syncthreads();
  BQ_PROF_MARK(2);
  // ---- exclusive scan of my 2048 counters (4 consecutive per thread), offset by the lower slabs' points
  int base = 0;
  for (int i = 0; i < kBuildThreads / kWave; ++i) base += s_red[0][i];
  const int4 cnt = *reinterpret_cast<const int4 *>(s_cnt + tid * kPerThread);
  static_assert(kPerThread == 4, "scan step below handles one int4 per thread");
  const int sum = cnt.x + cnt.y + cnt.z + cnt.w;
  int incl = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  if (lane == kWave - 1) s_red[1][wv] = incl;
  __syncthreads();
  int run = base + incl - sum;
  for (int i = 0; i < wv; ++i) run += s_red[1][i];
  int4 start;
  start.x = run; run += cnt.x;
  start.y = run; run += cnt.y;
  start.z = run; run += cnt.z;
  start.w = run; run += cnt.w;
  *reinterpret_cast<int4 *>(s_cnt + tid * kPerThread) = start;                   // scatter cursors
  *reinterpret_cast<int4 *>(w.cell_start + lo + tid * kPerThread) = start;
  if (slab == kSlabs - 1 && tid == kBuildThreads - 1) w.cell_start[kCells] = run;
  __syncthreads();
  BQ_PROF_MARK(3);
  // ---- scatter my points (order inside a cell is arbitrary; the query ranks hits by index)
  auto scatter_quad = [&](const Quad &qq, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = qq.v[3 * j], y = qq.v[3 * j + 1], z = qq.v[3 * j + 2];
      if (4 * g + j < n) {
        const int dz = fold_z(lattice(z, inv_cell)) - zlo;
        if (dz >= 0 && dz < kSlabLayers) {
          const int c = (dz * kGridY + fold_y(lattice(y, inv_cell))) * kGridX + fold_x(lattice(x, inv_cell));
          const int pos = atomicAdd(&s_cnt[c], 1);
          w.records[pos] = make_float4(x, y, z, __int_as_float(4 * g + j));
        }
      }
    }
  };
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) {
      // opaque copies: the cell index is RE-computed (a few VALU ops) instead of being carried across the
      // scan in a register per point -- registers are what limits the points a thread can keep
#pragma unroll
      for (int e = 0; e < 12; ++e) asm volatile("" : "+v"(q[i].v[e]));
      scatter_quad(q[i], tid + i * kBuildThreads);
    }
    for (int g = tid + GG * kBuildThreads; g < ngroups; g += kBuildThreads) scatter_quad(load_quad(pts, g, n, vec != 0), g);
  } else {
    for (int g = tid; g < ngroups; g += kBuildThreads) scatter_quad(load_quad(pts, g, n, vec != 0), g);
  }
  BQ_PROF_MARK(4);
  BQ_PROF_COUNT(9, 1);
#define BQ_PROF_WAVE wv
#define BQ_PROF_BASE 8192
  BQ_PROF_STORE;
#undef BQ_PROF_WAVE
#undef BQ_PROF_BASE
}

// ---- build, one workgroup per scene (round 5; large batches) -------------------------------------------------------
// The slab build above classifies every point in every one of a scene's 8 workgroups: 8x the work for 8x the CUs, which
// pays while there are fewer scenes than CUs / 8.  At the global batch (64 scenes = 512 slab workgroups = two rounds of
// the chip, 28 us) the redundancy is what the time is made of; here ONE 1024-thread workgroup per scene keeps the whole
// 16 384-counter table in LDS (64 KB) and every point is classified once per pass: count (LDS atomics), scan (16
// counters per thread), scatter (returning LDS atomics) -- the same table and records as the slab build (the order
// inside a cell is arbitrary in both; the query ranks by index).
constexpr int kBuild1Threads = 1024;
constexpr int kBuild1PerThread = kCells / kBuild1Threads;  // 16 counters per thread in the scan

template <int G>
__global__ __launch_bounds__(kBuild1Threads) void grid_build1_kernel(const float *__restrict__ xyz, int n, float inv_cell,
                                                                    unsigned char *__restrict__ ws, size_t scene_stride,
                                                                    int vec) {
  __shared__ __attribute__((aligned(16))) int s_cnt[kCells];
  __shared__ int s_wsum[kBuild1Threads / kWave];
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
  const int scene = blockIdx.x;
  const float *__restrict__ pts = xyz + static_cast<size_t>(scene) * n * 3;
  const SceneWs w = scene_ws(ws, scene_stride, scene);
  const int ngroups = (n + 3) / 4;

  for (int c = tid; c < kCells; c += kBuild1Threads) s_cnt[c] = 0;
  constexpr int GG = G > 0 ? G : 1;
  Quad q[GG];
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) q[i] = load_quad(pts, tid + i * kBuild1Threads, n, vec != 0);
  }
  __syncthreads();
  auto count_quad = [&](const Quad &qq, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * g + j < n) atomicAdd(&s_cnt[cell_of(qq.v[3 * j], qq.v[3 * j + 1], qq.v[3 * j + 2], inv_cell)], 1);
  };
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) count_quad(q[i], tid + i * kBuild1Threads);
    for (int g = tid + GG * kBuild1Threads; g < ngroups; g += kBuild1Threads) count_quad(load_quad(pts, g, n, vec != 0), g);
  } else {
    for (int g = tid; g < ngroups; g += kBuild1Threads) count_quad(load_quad(pts, g, n, vec != 0), g);
  }
  __syncthreads();
  // exclusive scan of the 16 384 counters: 16 consecutive ones per thread (read twice from LDS rather than kept: the
  // resident points need the registers)
  int sum = 0;
#pragma unroll
  for (int i = 0; i < kBuild1PerThread / 4; ++i) {
    const int4 c4 = *reinterpret_cast<const int4 *>(s_cnt + tid * kBuild1PerThread + 4 * i);
    sum += c4.x + c4.y + c4.z + c4.w;
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  if (lane == kWave - 1) s_wsum[wv] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int i = 0; i < wv; ++i) run += s_wsum[i];
#pragma unroll
  for (int i = 0; i < kBuild1PerThread / 4; ++i) {
    const int4 c4 = *reinterpret_cast<const int4 *>(s_cnt + tid * kBuild1PerThread + 4 * i);
    int4 st;
    st.x = run; run += c4.x;
    st.y = run; run += c4.y;
    st.z = run; run += c4.z;
    st.w = run; run += c4.w;
    *reinterpret_cast<int4 *>(s_cnt + tid * kBuild1PerThread + 4 * i) = st;                 // scatter cursors
    *reinterpret_cast<int4 *>(w.cell_start + tid * kBuild1PerThread + 4 * i) = st;
  }
  if (tid == kBuild1Threads - 1) w.cell_start[kCells] = run;
  __syncthreads();
  auto scatter_quad = [&](const Quad &qq, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = qq.v[3 * j], y = qq.v[3 * j + 1], z = qq.v[3 * j + 2];
      if (4 * g + j < n) {
        const int pos = atomicAdd(&s_cnt[cell_of(x, y, z, inv_cell)], 1);
        w.records[pos] = make_float4(x, y, z, __int_as_float(4 * g + j));
      }
    }
  };
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) {
#pragma unroll
      for (int e = 0; e < 12; ++e) asm volatile("" : "+v"(q[i].v[e]));  // (the cell index is recomputed, not carried)
      scatter_quad(q[i], tid + i * kBuild1Threads);
    }
    for (int g = tid + GG * kBuild1Threads; g < ngroups; g += kBuild1Threads) scatter_quad(load_quad(pts, g, n, vec != 0), g);
  } else {
    for (int g = tid; g < ngroups; g += kBuild1Threads) scatter_quad(load_quad(pts, g, n, vec != 0), g);
  }
}

// Keep only the `keep` smallest-index records of buf[0..h): all-pairs rank in LDS.
// Each lane owns records lane, lane+64, ...; returns the new count.
template <int kSlots>
__device__ __forceinline__ int rank_and_keep_n(float4 *buf, int h, int keep, int lane) {
  float4 mine[kSlots];
  int rank[kSlots];
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const int q = lane + j * kWave;
    mine[j] = q < h ? buf[q] : make_float4(0, 0, 0, __int_as_float(0x7fffffff));
    rank[j] = 0;
  }
  for (int t = 0; t < h; ++t) {
    const int v = __float_as_int(buf[t].w);  // wave-uniform LDS broadcast
#pragma unroll
    for (int j = 0; j < kSlots; ++j) rank[j] += (v < __float_as_int(mine[j].w)) ? 1 : 0;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const int q = lane + j * kWave;
    if (q < h && rank[j] < keep) buf[rank[j]] = mine[j];  // indices are unique -> ranks are a permutation
  }
  __builtin_amdgcn_wave_barrier();
  return h < keep ? h : keep;
}

__device__ __forceinline__ int rank_and_keep(float4 *buf, int h, int keep, int lane) {
  if (h <= kWave) return rank_and_keep_n<1>(buf, h, keep, lane);
  return rank_and_keep_n<kHitCap / kWave>(buf, h, keep, lane);
}

template <int DM>
__global__ __launch_bounds__(kQueryWaves *kWave) void grid_query_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int n,
    unsigned char *__restrict__ ws, size_t scene_stride, int32_t *__restrict__ idx,
    float *__restrict__ grouped, int m, float r2, float inv_radius, float inv_cell, int nsample, int normalize,
    int nscenes) {
  __shared__ float4 s_hits[kQueryWaves][kHitCap];

  const int w = wave_id();
  const int lane = lane_id();
  // XCD-aware mapping: workgroup id g runs on XCD g % 8 (observed dispatch order), so with
  // scene = g % B all workgroups of a scene share ONE XCD's L2 (for B = 8: scene s <-> XCD s)
  // and a scene's records are fetched from HBM once instead of once per XCD.  Only speed
  // depends on the placement.
  const int bi = blockIdx.x % nscenes;
  const int j = (blockIdx.x / nscenes) * kQueryWaves + w;
  if (j >= m) return;  // wave-uniform, no workgroup barrier in this kernel

  const SceneWs sw = scene_ws(ws, scene_stride, bi);
  const int *__restrict__ cell_start = sw.cell_start;
  const float4 *__restrict__ records = sw.records;
  float4 *buf = s_hits[w];

  const float *ctr = new_xyz + (static_cast<size_t>(bi) * m + j) * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int icx = lattice(cx, inv_cell), icy = lattice(cy, inv_cell), icz = lattice(cz, inv_cell);

  // nine (dy, dz) rows of three x-adjacent cells.  On the torus the three cells are memory-contiguous
  // unless they straddle the row end: lanes 0..8 hold the part up to the row end, lanes 9..17 the
  // wrapped remainder (empty, and not visited, when xs <= 29 -- wave-uniform).
  const int xs = fold_x(icx - 1);
  const int first = min(3, kGridX - xs);
  const int nranges = first < 3 ? 18 : 9;
  int start = 0, len = 0;
  if (lane < nranges) {
    const int row = lane < 9 ? lane : lane - 9;
    const int iy = fold_y(icy + (row % 3) - 1), iz = fold_z(icz + (row / 3) - 1);
    const int rowbase = (iz * kGridY + iy) * kGridX;
    const int c0 = lane < 9 ? rowbase + xs : rowbase;
    const int ncell = lane < 9 ? first : 3 - first;
    start = cell_start[c0];
    len = cell_start[c0 + ncell] - start;
  }
  int incl = len;
  for (int off = 1; off < 32; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const int total = __builtin_amdgcn_readlane(incl, 17);

  const uint64_t below = (1ull << lane) - 1ull;
  int nhits = 0;
  // Candidates are fetched kUn x 64 at a time: the record gathers of a whole batch (random 16-byte reads, an L2 round
  // trip each) are in flight together instead of one dependent round trip per 64 candidates -- a ball has 130-300
  // candidates, so most centres need ONE batch.  The range search is shared by the batch's kUn positions.
  constexpr int kUn = 4;
  for (int t0 = 0; t0 < total; t0 += kWave * kUn) {
    int src[kUn];
#pragma unroll
    for (int u = 0; u < kUn; ++u) src[u] = -1;
    for (int r = 0; r < nranges; ++r) {
      const int r_incl = __builtin_amdgcn_readlane(incl, r);
      const int r_len = __builtin_amdgcn_readlane(len, r);
      const int r_first = __builtin_amdgcn_readlane(start, r) - (r_incl - r_len);
#pragma unroll
      for (int u = 0; u < kUn; ++u) {
        const int t = t0 + u * kWave + lane;
        if (src[u] < 0 && t < r_incl) src[u] = r_first + t;
      }
    }
    float4 p[kUn];
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      const bool valid = t0 + u * kWave + lane < total;
      p[u] = valid ? records[src[u]] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < kUn; ++u) {
      if (t0 + u * kWave >= total) break;  // wave-uniform
      const bool valid = t0 + u * kWave + lane < total;
      const float d2 = sqdist3<DM>(__fsub_rn(cx, p[u].x), __fsub_rn(cy, p[u].y), __fsub_rn(cz, p[u].z));
      const bool hit = valid && d2 < r2;  // same fp32 expression as the scan: ball_query_gpu.cu:34-36
      const uint64_t mask = __ballot(hit);
      const int add = __popcll(mask);
      if (add) {
        if (nhits + add > kHitCap) nhits = rank_and_keep(buf, nhits, nsample, lane);
        if (hit) buf[nhits + __popcll(mask & below)] = p[u];
        nhits += add;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  nhits = rank_and_keep(buf, nhits, nsample, lane);  // buf[0..nhits) ascending by point index

  const size_t row_off = (static_cast<size_t>(bi) * m + j) * nsample;
  const size_t plane = static_cast<size_t>(m) * nsample;
  for (int s = lane; s < nsample; s += kWave) {
    // pad with the first hit (:37-41); an empty ball keeps index 0 (zero-filled output)
    float4 p;
    if (nhits > 0) {
      p = buf[s < nhits ? s : 0];
    } else {  // the reference then groups point 0 of the scene
      const float *p0 = xyz + static_cast<size_t>(bi) * n * 3;
      p = make_float4(p0[0], p0[1], p0[2], __int_as_float(0));
    }
    const int v = __float_as_int(p.w);
    idx[row_off + s] = v;
    if (grouped) {
      float gx = __fsub_rn(p.x, cx), gy = __fsub_rn(p.y, cy), gz = __fsub_rn(p.z, cz);
      if (normalize & 1) {
        gx = __fmul_rn(gx, inv_radius); gy = __fmul_rn(gy, inv_radius); gz = __fmul_rn(gz, inv_radius);
      }
      if (normalize & 2) {  // channels-last (B,M,S,3)
        float *g = grouped + (row_off + s) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
      } else {
        float *g = grouped + static_cast<size_t>(bi) * 3 * plane + static_cast<size_t>(j) * nsample + s;
        g[0] = gx;
        g[plane] = gy;
        g[2 * plane] = gz;
      }
    }
  }
}


// ---- query, round 5: EIGHT LANES PER CENTRE ---------------------------------------------------------------------
// The one-wave-per-centre kernel above executes ~500-900 instructions per centre with most lanes idle most of the time
// (18 of 64 lanes fetch cell ranges, the range search is redone per chunk, the rank loop walks the hits one LDS
// broadcast at a time); both at the global batch (131 072 centres) and at 8 scenes the operator is bound by that
// instruction count, not by bytes (tools/bq_phase.py: phase by phase).  Here a wave serves EIGHT centres, eight lanes
// each:
//  * lanes 0..7 of a centre fetch the nine (dy, dz) cell rows (lane 0 two of them), both pieces of a row that wraps
//    around the torus; an 8-lane scan turns them into a table of 18 (end, record offset) pieces in LDS -- the ball's
//    candidates as ONE flattened list;
//  * lane k walks candidates k, k + 8, ... of that list (advancing through the table as it goes), four gathers in
//    flight per lane and the next four issued before the current four are tested;
//  * hits are compacted per 8-lane group with a ballot slice + popcount into the centre's LDS list (point indices
//    only), the running length lives in a register;
//  * the "first nsample in index order" rule is a bitonic SORTING NETWORK on registers: 64 indices per centre, eight
//    per lane, the six cross-lane stages as DPP quad_perm / row_half_mirror operands, the fifteen in-lane stages as
//    plain min / max pairs -- ~290 instructions for the eight centres of a wave, whatever the list lengths.  A list
//    that outgrows its LDS capacity is cut to its 64 smallest on the way (sort both halves, elementwise min against
//    the mirrored other half, one bitonic merge), which later hits can only displace larger indices from;
//  * the survivors' coordinates are gathered from the cloud by index and written as 16-byte stores.
// nsample <= 64 (the reference's callers use 64 and 32); larger nsample keeps the one-wave-per-centre kernel.
// Output identical to the serial scan (ball_query_gpu.cu:26-46).
constexpr int kQ8Lanes = 8;                        // lanes per centre
constexpr int kQ8PerWave = kWave / kQ8Lanes;       // 8 centres per wave
constexpr int kQ8Waves = 4;
constexpr int kQ8Centres = kQ8Waves * kQ8PerWave;  // 32 per workgroup
constexpr int kQ8Pieces = 18;                      // 9 rows x (piece up to the row end, wrapped piece)
constexpr int kQ8Cap = 128;                        // list entries per centre in LDS
constexpr int kQ8Unroll = 4;                       // candidates per lane and step
constexpr int kQ8MaxSample = 64;                   // what one sorting network holds
constexpr int kIntMax = 0x7fffffff;

// dpp_ctrl encodings (gfx9): quad_perm [a,b,c,d] = a | b << 2 | c << 4 | d << 6; row_half_mirror = 0x141
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppXor3 = 0x1B, kDppHalfMirror = 0x141;
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// value of lane (k ^ LM) of the same 8-lane group, LM in {1, 2, 3, 4, 7}
template <int LM>
__device__ __forceinline__ int group_xor(int v) {
  if constexpr (LM == 1) return dpp_i32<kDppXor1>(v);
  else if constexpr (LM == 2) return dpp_i32<kDppXor2>(v);
  else if constexpr (LM == 3) return dpp_i32<kDppXor3>(v);
  else if constexpr (LM == 7) return dpp_i32<kDppHalfMirror>(v);
  else return dpp_i32<kDppXor3>(dpp_i32<kDppHalfMirror>(v));  // k -> 7 - (k ^ 3) = k ^ 4
}
// Element e = 8 * k + r of a centre's 64 lives in register r of lane k (k = lane within the 8-lane group).
// One compare-exchange stage between e and e ^ M for M < 8 (xor stage) or the in-lane "flip" of a merge's first stage:
// both pair register r with r ^ M and leave the minimum in the lower one.
template <int M>
__device__ __forceinline__ void cx_inlane(int (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if ((r ^ M) > r) {
      const int lo = min(v[r], v[r ^ M]), hi = max(v[r], v[r ^ M]);
      v[r] = lo;
      v[r ^ M] = hi;
    }
  }
}
// ... and between e and e ^ (8 * LM + RM): the partner is register r ^ RM of lane k ^ LM; the element whose lane has
// the top bit of LM clear is the lower one and keeps the minimum.
template <int LM, int RM>
__device__ __forceinline__ void cx_cross(int (&v)[8], int k) {
  constexpr int top = LM >= 4 ? 4 : (LM >= 2 ? 2 : 1);
  int p[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) p[r] = group_xor<LM>(v[r ^ RM]);
  const bool lower = (k & top) == 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = lower ? min(v[r], p[r]) : max(v[r], p[r]);
}
// ascending sort of the 64 elements of every 8-lane group (all lanes of the wave take part): the bitonic network in
// its "flip" form (first stage of a merge pairs e with e ^ (size - 1)), so every stage orders towards the lower index
__device__ __forceinline__ void sort64(int (&v)[8], int k) {
  cx_inlane<1>(v);
  cx_inlane<3>(v); cx_inlane<1>(v);
  cx_inlane<7>(v); cx_inlane<2>(v); cx_inlane<1>(v);
  cx_cross<1, 7>(v, k); cx_inlane<4>(v); cx_inlane<2>(v); cx_inlane<1>(v);
  cx_cross<3, 7>(v, k); cx_cross<1, 0>(v, k); cx_inlane<4>(v); cx_inlane<2>(v); cx_inlane<1>(v);
  cx_cross<7, 7>(v, k); cx_cross<2, 0>(v, k); cx_cross<1, 0>(v, k); cx_inlane<4>(v); cx_inlane<2>(v); cx_inlane<1>(v);
}
// a, b ascending -> a = the 64 smallest of both, ascending (min against the mirrored b is bitonic; one merge sorts it)
__device__ __forceinline__ void merge_low64(int (&a)[8], const int (&b)[8], int k) {
#pragma unroll
  for (int r = 0; r < 8; ++r) a[r] = min(a[r], group_xor<7>(b[r ^ 7]));
  cx_cross<4, 0>(a, k); cx_cross<2, 0>(a, k); cx_cross<1, 0>(a, k); cx_inlane<4>(a); cx_inlane<2>(a); cx_inlane<1>(a);
}
__device__ __forceinline__ int group_max8(int v) {  // v uniform per 8-lane group -> the maximum over the wave's groups
  int mx = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int q = 1; q < kQ8PerWave; ++q) mx = max(mx, __builtin_amdgcn_readlane(v, q * kQ8Lanes));
  return mx;
}
// The 64 smallest entries of list[0..cnt) (cnt <= kQ8Cap, uniform per group), ascending, padded with INT_MAX, into v.
__device__ __forceinline__ void sorted_low64(const int *list, int cnt, int k, int (&v)[8]) {
  const int4 a = *reinterpret_cast<const int4 *>(list + 8 * k), b = *reinterpret_cast<const int4 *>(list + 8 * k + 4);
  const int raw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = 8 * k + r < cnt ? raw[r] : kIntMax;
  sort64(v, k);
  if (group_max8(cnt) > 64) {  // wave-uniform
    const int4 c = *reinterpret_cast<const int4 *>(list + 64 + 8 * k), d = *reinterpret_cast<const int4 *>(list + 68 + 8 * k);
    const int raw2[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    int x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = 64 + 8 * k + r < cnt ? raw2[r] : kIntMax;
    sort64(x, k);
    merge_low64(v, x, k);
  }
}

template <int DM>
__global__ __launch_bounds__(kQ8Waves *kWave, 3) void grid_query8_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int n, unsigned char *__restrict__ ws,
    size_t scene_stride, int32_t *__restrict__ idx, float *__restrict__ grouped, int m, float r2, float inv_radius,
    float inv_cell, int nsample, int normalize, int nscenes, int vec_out) {
  // (the list's 128 entries + 64 more floats: the channels-last output row is transposed through these 768 bytes)
  __shared__ __attribute__((aligned(16))) int s_list[kQ8Centres][kQ8Cap + 64];
  __shared__ __attribute__((aligned(8))) int2 s_piece[kQ8Centres][kQ8Pieces + 2];  // (end in the flattened list, record - t) of the non-empty pieces + sentinels

  const int w = wave_id(), lane = lane_id();
  const int g = lane >> 3, k = lane & (kQ8Lanes - 1);
  const int slot = w * kQ8PerWave + g;
  const int bi = blockIdx.x % nscenes;  // a scene's workgroups share one XCD's L2 (see grid_query_kernel)
  const int jraw = (blockIdx.x / nscenes) * kQ8Centres + slot;
  const bool live = jraw < m;            // groups past the last centre repeat it and write nothing
  const int j = live ? jraw : m - 1;

  const SceneWs sw = scene_ws(ws, scene_stride, bi);
  const int *__restrict__ cell_start = sw.cell_start;
  const float4 *__restrict__ records = sw.records;
  int *list = s_list[slot];
  int2 *piece = s_piece[slot];

  BQ_PROF_DECL;
  const float *ctr = new_xyz + (static_cast<size_t>(bi) * m + j) * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int icx = lattice(cx, inv_cell), icy = lattice(cy, inv_cell), icz = lattice(cz, inv_cell);
  const int xs = fold_x(icx - 1);
  const int first = min(3, kGridX - xs);  // x cells up to the row end; the rest (if any) wraps to the row start
#ifdef CODA_BQ_PROF
  asm volatile("" ::"v"(cx), "v"(cy), "v"(cz));
#endif
  BQ_PROF_MARK(0);
  // ---- lane k: row k (lane 0 also row 8), as (start, length) of the piece up to the row end and of the wrapped piece
  int a0, n0, a1, n1, b0 = 0, m0 = 0, b1 = 0, m1 = 0;
  {
    const int iy = fold_y(icy + (k % 3) - 1), iz = fold_z(icz + (k / 3) - 1);
    const int rowbase = (iz * kGridY + iy) * kGridX;
    a0 = cell_start[rowbase + xs];
    n0 = cell_start[rowbase + xs + first] - a0;
    a1 = cell_start[rowbase];                        // (read unconditionally: no divergent branch in front of the loads)
    n1 = first < 3 ? cell_start[rowbase + 3 - first] - a1 : 0;
  }
  if (k == 0) {  // row 8: dy = +1, dz = +1
    const int rowbase = (fold_z(icz + 1) * kGridY + fold_y(icy + 1)) * kGridX;
    b0 = cell_start[rowbase + xs];
    m0 = cell_start[rowbase + xs + first] - b0;
    b1 = cell_start[rowbase];
    m1 = first < 3 ? cell_start[rowbase + 3 - first] - b1 : 0;
  }
  // 8-lane scan of (candidates, non-empty pieces) of rows 0..7 in one word; row 8 goes last.  Only NON-EMPTY pieces
  // enter the table (a ball on a wall has three long rows and six empty ones, and the wrapped pieces are nearly always
  // empty): the walk below reads the table serially, one LDS round trip per piece it steps over.
  const int len = n0 + n1;
  const int npc = (n0 > 0 ? 1 : 0) + (n1 > 0 ? 1 : 0);
  int incl = (len << 5) | npc;
#pragma unroll
  for (int d = 1; d < kQ8Lanes; d <<= 1) {
    const int up = __shfl_up(incl, d, kQ8Lanes);
    if (k >= d) incl += up;
  }
  const int excl = (incl >> 5) - len;          // candidates in front of this lane's row
  int at = (incl & 31) - npc;                  // table entries in front of it
  const int last = __shfl(incl, kQ8Lanes - 1, kQ8Lanes);
  const int t8 = last >> 5, np8 = last & 31;   // rows 0..7 together
  const int len8 = __shfl(m0 + m1, 0, kQ8Lanes);
  const int total = t8 + len8;                 // candidates of this centre
  if (n0 > 0) piece[at++] = make_int2(excl + n0, a0 - excl);
  if (n1 > 0) piece[at] = make_int2(excl + len, a1 - (excl + n0));
  if (k == 0) {
    int e = np8;
    if (m0 > 0) piece[e++] = make_int2(t8 + m0, b0 - t8);
    if (m1 > 0) piece[e++] = make_int2(t8 + m0 + m1, b1 - (t8 + m0));
    piece[e] = make_int2(kIntMax, 0);  // sentinel: the walk stops here; candidates past the end read records[t]
    piece[e + 1] = make_int2(kIntMax, 0);
  }
  __builtin_amdgcn_wave_barrier();
  BQ_PROF_MARK(1);

  // ---- walk the flattened list: lane k takes candidates k, k + 8, ...; the gathers of the next TWO steps are in flight
  // while a step is tested
  const int steps = group_max8((total + kQ8Lanes - 1) / kQ8Lanes);  // per lane, the longest list of the wave
  int cnt = 0;       // entries in this centre's list (uniform per group)
  int pi = 0;        // table entry the walk is in; the next one is kept in registers (its LDS latency is off the path)
  int2 pc = piece[0], pn = piece[1];
  // three register sets of four candidates in rotation (no copies: a copy of a register whose load is still in flight
  // waits for it, which is what made the first version of this loop pay a full memory round trip per step)
  float4 ra[kQ8Unroll], rb[kQ8Unroll], rc[kQ8Unroll];
  unsigned int va = 0u, vb = 0u, vc = 0u;  // bit u: candidate u of the set exists
  auto fetch = [&](int i0, float4 (&rec)[kQ8Unroll], unsigned int &val) {
    val = 0u;
#pragma unroll
    for (int u = 0; u < kQ8Unroll; ++u) {
      const int t = k + kQ8Lanes * (i0 + u);
      while (t >= pc.x) {
        pc = pn;
        pn = piece[++pi + 1];
      }
      val |= (t < total ? 1u : 0u) << u;
      // past the end: a low record of the scene, so that the idle lanes' loads neither fault nor pile onto one address
      rec[u] = records[min(pc.y + t, n - 1)];
    }
  };
  // my byte of a ballot: lanes 8 g .. 8 g + 7
  const bool upper_half = g >= 4;
  const int byte_shift = (g & 3) * 8;
  const unsigned int below8 = (1u << k) - 1u;
  auto consume = [&](const float4 (&rec)[kQ8Unroll], unsigned int val) {
    if (__ballot(cnt > kQ8Cap - kQ8Unroll * kQ8Lanes) != 0ull) {
      // a list could overflow in this step: every list of the wave down to its 64 smallest (wave-uniform branch)
      int v[8];
      sorted_low64(list, cnt, k, v);
      __builtin_amdgcn_wave_barrier();
      *reinterpret_cast<int4 *>(list + 8 * k) = make_int4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<int4 *>(list + 8 * k + 4) = make_int4(v[4], v[5], v[6], v[7]);
      cnt = min(cnt, 64);
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int u = 0; u < kQ8Unroll; ++u) {
      const float4 c = rec[u];
      const float d2 = sqdist3<DM>(__fsub_rn(cx, c.x), __fsub_rn(cy, c.y), __fsub_rn(cz, c.z));
      const bool hit = ((val >> u) & 1u) != 0u && d2 < r2;  // the scan's fp32 expression: ball_query_gpu.cu:34-36
      const uint64_t mask = __ballot(hit);
      const unsigned int half = upper_half ? static_cast<unsigned int>(mask >> 32) : static_cast<unsigned int>(mask);
      const unsigned int mine = (half >> byte_shift) & 0xffu;
      if (hit) list[cnt + __popc(mine & below8)] = __float_as_int(c.w);
      cnt += __popc(mine);
    }
  };
  fetch(0, ra, va);
  fetch(kQ8Unroll, rb, vb);
  BQ_PROF_MARK(2);
  BQ_PROF_COUNT(8, steps);
  BQ_PROF_COUNT(9, 1);
  for (int i0 = 0; i0 < steps; i0 += 3 * kQ8Unroll) {  // (the fetches are unconditional: the waits in front of the tests stay exact)
    fetch(i0 + 2 * kQ8Unroll, rc, vc);
    consume(ra, va);
    if (i0 + kQ8Unroll >= steps) break;
    fetch(i0 + 3 * kQ8Unroll, ra, va);
    consume(rb, vb);
    if (i0 + 2 * kQ8Unroll >= steps) break;
    fetch(i0 + 4 * kQ8Unroll, rb, vb);
    consume(rc, vc);
  }
  __builtin_amdgcn_wave_barrier();
  BQ_PROF_MARK(3);
  // ---- the nsample smallest point indices, ascending
  {
    int v[8];
    sorted_low64(list, cnt, k, v);
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<int4 *>(list + 8 * k) = make_int4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<int4 *>(list + 8 * k + 4) = make_int4(v[4], v[5], v[6], v[7]);
    __builtin_amdgcn_wave_barrier();
  }
  const int nh = min(cnt, nsample);
  BQ_PROF_MARK(4);
  BQ_PROF_COUNT(10, cnt);

#ifndef CODA_BQ_PROF
  if (!live) return;
#endif
  const size_t row_off = (static_cast<size_t>(bi) * m + j) * nsample;
  const size_t plane = static_cast<size_t>(m) * nsample;
  const float *p0 = xyz + static_cast<size_t>(bi) * n * 3;
  const int head = nh > 0 ? list[0] : 0;  // pad with the first hit (:37-41); an empty ball keeps index 0 (zero-filled
                                          // output) and the reference then groups point 0 of the scene
  auto centred = [&](int i, float &gx, float &gy, float &gz) {
    const float *q = p0 + static_cast<size_t>(i) * 3;
    gx = __fsub_rn(q[0], cx); gy = __fsub_rn(q[1], cy); gz = __fsub_rn(q[2], cz);
    if (normalize & 1) {
      gx = __fmul_rn(gx, inv_radius); gy = __fmul_rn(gy, inv_radius); gz = __fmul_rn(gz, inv_radius);
    }
  };
  if (vec_out) {  // nsample % 4 == 0 and 16-byte aligned outputs: four samples per lane and store
    // all of the row's indices into registers first: the list's LDS doubles as the transposition buffer below
    int4 raws[kQ8MaxSample / (4 * kQ8Lanes)];
#pragma unroll
    for (int it = 0; it < kQ8MaxSample / (4 * kQ8Lanes); ++it)
      raws[it] = *reinterpret_cast<const int4 *>(list + 4 * k + it * 4 * kQ8Lanes);
    __builtin_amdgcn_wave_barrier();
    float *stage = reinterpret_cast<float *>(list);
#pragma unroll
    for (int it = 0; it < kQ8MaxSample / (4 * kQ8Lanes); ++it) {
      if (it * 4 * kQ8Lanes >= nsample) break;  // (uniform: the whole pass lies past the row)
      const int s4 = 4 * k + it * 4 * kQ8Lanes;
      // A lane whose four samples lie past the row (nsample = 4, 8, 16, 24, 48 ...) has nothing of its own but STAYS in
      // the pass: the channels-last store below is cooperative (lane k writes pieces k, k + 8, k + 16 of the group's
      // staged rows).  Round 5 left the loop here instead, and pieces 4..7 of a 16-sample row were never written.
      const bool mine = s4 < nsample;
      const int4 raw = raws[it];
      const int i4[4] = {s4 < nh ? raw.x : head, s4 + 1 < nh ? raw.y : head, s4 + 2 < nh ? raw.z : head,
                         s4 + 3 < nh ? raw.w : head};
      if (mine) *reinterpret_cast<int4 *>(idx + row_off + s4) = make_int4(i4[0], i4[1], i4[2], i4[3]);
      if (grouped) {
        float gx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gy[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (mine) {
#pragma unroll
          for (int q = 0; q < 4; ++q) centred(i4[q], gx[q], gy[q], gz[q]);
        }
        if (normalize & 2) {  // channels-last (B,M,S,3): 12 consecutive floats per lane, 96 per group and pass
          // Through LDS, so that every store instruction writes whole lines: lane k's three 16-byte pieces are 48 bytes
          // apart -- stored directly, each instruction wrote every third 16-byte piece of its lines (PMC: 22.0 MB
          // written and 7 MB fetched back for 16.8 MB of rows).  Transposed, lane k writes pieces k, k + 8, k + 16.
          if (mine) {
            float4 *st = reinterpret_cast<float4 *>(stage + 3 * s4);
            st[0] = make_float4(gx[0], gy[0], gz[0], gx[1]);
            st[1] = make_float4(gy[1], gz[1], gx[2], gy[2]);
            st[2] = make_float4(gz[2], gx[3], gy[3], gz[3]);
          }
          __builtin_amdgcn_wave_barrier();
          const int base4 = it * 3 * kQ8Lanes;  // 16-byte pieces of this pass start here
          const float4 *rd = reinterpret_cast<const float4 *>(stage);
          float4 *o = reinterpret_cast<float4 *>(grouped + row_off * 3);
          const int npieces = min(nsample - it * 4 * kQ8Lanes, 4 * kQ8Lanes) * 3 / 4;  // of this pass (nsample % 4 == 0)
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int pc4 = k + kQ8Lanes * r;
            if (pc4 < npieces) o[base4 + pc4] = rd[base4 + pc4];
          }
        } else if (mine) {
          float *o = grouped + static_cast<size_t>(bi) * 3 * plane + static_cast<size_t>(j) * nsample + s4;
          *reinterpret_cast<float4 *>(o) = make_float4(gx[0], gx[1], gx[2], gx[3]);
          *reinterpret_cast<float4 *>(o + plane) = make_float4(gy[0], gy[1], gy[2], gy[3]);
          *reinterpret_cast<float4 *>(o + 2 * plane) = make_float4(gz[0], gz[1], gz[2], gz[3]);
        }
      }
    }
  } else {
    for (int sidx = k; sidx < nsample; sidx += kQ8Lanes) {
      const int i = sidx < nh ? list[sidx] : head;
      idx[row_off + sidx] = i;
      if (grouped) {
        float gx, gy, gz;
        centred(i, gx, gy, gz);
        if (normalize & 2) {
          float *o = grouped + (row_off + sidx) * 3;
          o[0] = gx; o[1] = gy; o[2] = gz;
        } else {
          float *o = grouped + static_cast<size_t>(bi) * 3 * plane + static_cast<size_t>(j) * nsample + sidx;
          o[0] = gx;
          o[plane] = gy;
          o[2 * plane] = gz;
        }
      }
    }
  }
  BQ_PROF_MARK(5);
#define BQ_PROF_WAVE w
#define BQ_PROF_BASE 0
  BQ_PROF_STORE;
#undef BQ_PROF_WAVE
#undef BQ_PROF_BASE
}

}  // namespace

// Entry used by ball_query.hip's dispatcher.
int ball_query_grid(const float *new_xyz, const float *xyz, int32_t *idx, float *grouped, int b, int n,
                    int m, float radius, int nsample, int normalize, void *workspace, hipStream_t s) {
  const size_t stride = (grid_scene_bytes(n) + 255) & ~static_cast<size_t>(255);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  const float cell = fmaxf(radius * 1.001f, 1e-30f);
  const float inv_cell = 1.0f / cell;
  clear_sticky_error();
  // 16-B loads of four points need a 16-B aligned scene base: n % 4 == 0 (and an aligned tensor)
  const int vec = (n % 4 == 0 && (reinterpret_cast<uintptr_t>(xyz) & 15) == 0) ? 1 : 0;
  const int per = ceil_div(ceil_div(n, 4), kBuildThreads);  // groups of four points per thread
  const dim3 bgrid(b * kSlabs);
  // G = quads a thread keeps in registers across the count and the scatter pass.  Round 2 kept 12 for a cloud of
  // 20 000 points (10 needed): that instantiation spills (164 B of scratch per lane = 5.4 MB of HBM traffic per call
  // at B = 8 -- what the PMC table of profiles/r02_pmc_ball_query.md showed as "partial-line writes").  G = 8 is the
  // largest spill-free one; the groups beyond it are read again from L2 in the second pass.
  // CODA_BQ_KEEP=0|2|6|8|12 forces an instantiation (A/B).
  static const int force = [] { const char *e = getenv("CODA_BQ_KEEP"); return e ? atoi(e) : -1; }();
  const int keep = force >= 0 ? force : (per <= 2 ? 2 : (per <= 6 ? 6 : 8));
  // From CODA_BQ_BUILD1 scenes on (default 24: where the slab build's b * 8 workgroups stop fitting the chip in one
  // round at two per CU... measured: see profiles/r05_pmc_ball_query.md) ONE workgroup per scene builds the whole table.
  static const int build1_from = [] { const char *e = getenv("CODA_BQ_BUILD1"); return e ? atoi(e) : 24; }();
  if (b >= build1_from) {
    const int q1 = ceil_div(ceil_div(n, 4), kBuild1Threads);  // quads per thread
    const dim3 g1(b), b1(kBuild1Threads);
    // four quads per thread stay in registers across the two passes (128 VGPRs per thread at 1024 threads: five spill),
    // the rest of the cloud is read again from L2 in the second pass
    (void)q1;
    hipLaunchKernelGGL(grid_build1_kernel<4>, g1, b1, 0, s, xyz, n, inv_cell, ws, stride, vec);
  } else if (keep == 2) hipLaunchKernelGGL(grid_build_kernel<2>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else if (keep == 6) hipLaunchKernelGGL(grid_build_kernel<6>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else if (keep == 8) hipLaunchKernelGGL(grid_build_kernel<8>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else if (keep == 12) hipLaunchKernelGGL(grid_build_kernel<kKeepMax>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else hipLaunchKernelGGL(grid_build_kernel<0>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  const float r2 = radius * radius;
  // CODA_BQ_QUERY=wave: round 2-4's one-wave-per-centre query for every nsample (A/B); default: eight centres per
  // wave where one sorting network holds the row (nsample <= 64)
  static const bool per_wave = [] { const char *e = getenv("CODA_BQ_QUERY"); return e && e[0] == 'w'; }();
  if (per_wave || nsample > kQ8MaxSample) {
    CODA_DISPATCH_DM(distance_mode(),
                     hipLaunchKernelGGL(grid_query_kernel<DM>, dim3(ceil_div(m, kQueryWaves) * b),
                                        dim3(kQueryWaves * kWave), 0, s, new_xyz, xyz, n, ws, stride, idx, grouped, m,
                                        r2, 1.0f / radius, inv_cell, nsample, normalize, b));
    return launch_status();
  }
  const int vec_out = (nsample % 4 == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0 &&
                       (!grouped || (reinterpret_cast<uintptr_t>(grouped) & 15) == 0)) ? 1 : 0;
  CODA_DISPATCH_DM(distance_mode(),
                   hipLaunchKernelGGL(grid_query8_kernel<DM>, dim3(ceil_div(m, kQ8Centres) * b), dim3(kQ8Waves * kWave), 0, s,
                                      new_xyz, xyz, n, ws, stride, idx, grouped, m, r2, 1.0f / radius, inv_cell, nsample,
                                      normalize, b, vec_out));
  return launch_status();
}

#ifdef CODA_BQ_PROF
extern "C" __attribute__((visibility("default"))) int coda_bq_prof_read(unsigned long long *host, int reset) {
  int st = static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bq_prof), sizeof(g_bq_prof)));
  if (reset) {
    void *dev = nullptr;
    st |= static_cast<int>(hipGetSymbolAddress(&dev, HIP_SYMBOL(g_bq_prof)));
    st |= static_cast<int>(hipMemset(dev, 0, sizeof(g_bq_prof)));
  }
  return st;
}
#endif

size_t ball_query_grid_workspace(int b, int n, int nsample) {
  if (n < kGridMinPoints || nsample > kGridMaxSample) return 0;
  const size_t stride = (grid_scene_bytes(n) + 255) & ~static_cast<size_t>(255);
  return stride * static_cast<size_t>(b);
}

}  // namespace coda
