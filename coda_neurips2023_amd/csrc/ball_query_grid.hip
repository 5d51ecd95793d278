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
// Query (one wave per centre): lanes 0..8 fetch the nine x-contiguous cell ranges
//   (18 half-ranges when the three x cells wrap around the torus), a wave prefix sum
//   flattens them into one candidate list so that every lane of every 64-wide chunk has
//   work, hits are compacted with ballot/mbcnt into an LDS buffer, ranked, and written as
//   one 256-B row (+ the centred / normalised xyz of the fused QueryAndGroup path).
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

  for (int c = tid; c < kSlabCells; c += kBuildThreads) s_cnt[c] = 0;
  constexpr int GG = G > 0 ? G : 1;
  Quad q[GG];
  if (G > 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i) q[i] = load_quad(pts, tid + i * kBuildThreads, n, vec != 0);
  }
  __syncthreads();
  // ---- count my cells; count the points of lower slabs
  int below = 0;
  auto count_quad = [&](const Quad &qq, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = qq.v[3 * j], y = qq.v[3 * j + 1], z = qq.v[3 * j + 2];
      if (4 * g + j < n) {
        const int c = cell_of(x, y, z, inv_cell) - lo;
        if (c < 0) ++below;
        else if (c < kSlabCells) atomicAdd(&s_cnt[c], 1);
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
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off);
  if (lane == 0) s_red[0][wv] = below;
  __syncthreads();
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
  // ---- scatter my points (order inside a cell is arbitrary; the query ranks hits by index)
  auto scatter_quad = [&](const Quad &qq, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = qq.v[3 * j], y = qq.v[3 * j + 1], z = qq.v[3 * j + 2];
      if (4 * g + j < n) {
        const int c = cell_of(x, y, z, inv_cell) - lo;
        if (c >= 0 && c < kSlabCells) {
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
  if (keep == 2) hipLaunchKernelGGL(grid_build_kernel<2>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else if (keep == 6) hipLaunchKernelGGL(grid_build_kernel<6>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else if (keep == 8) hipLaunchKernelGGL(grid_build_kernel<8>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else if (keep == 12) hipLaunchKernelGGL(grid_build_kernel<kKeepMax>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  else hipLaunchKernelGGL(grid_build_kernel<0>, bgrid, dim3(kBuildThreads), 0, s, xyz, n, inv_cell, ws, stride, vec, b);
  const float r2 = radius * radius;
  CODA_DISPATCH_DM(distance_mode(),
                   hipLaunchKernelGGL(grid_query_kernel<DM>, dim3(ceil_div(m, kQueryWaves) * b),
                                      dim3(kQueryWaves * kWave), 0, s, new_xyz, xyz, n, ws, stride, idx, grouped, m,
                                      r2, 1.0f / radius, inv_cell, nsample, normalize, b));
  return launch_status();
}

size_t ball_query_grid_workspace(int b, int n, int nsample) {
  if (n < kGridMinPoints || nsample > kGridMaxSample) return 0;
  const size_t stride = (grid_scene_bytes(n) + 255) & ~static_cast<size_t>(255);
  return stride * static_cast<size_t>(b);
}

}  // namespace coda
