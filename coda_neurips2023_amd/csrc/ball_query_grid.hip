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
// Kernel A (one 1024-thread workgroup per scene): bounding box of the finite
//   points, cell histogram with LDS atomics, exclusive scan, scatter of
//   (x, y, z, index) records into cell order.  Workspace layout per scene:
//   GridHeader | cell_start[kCellMax + 1] | records[N] (float4).
// Kernel B (one wave per centre): lanes 0..8 fetch the nine x-contiguous cell
//   ranges, a wave prefix sum flattens them into one candidate list so that
//   every lane of every 64-wide chunk has work, hits are compacted with
//   ballot/mbcnt into an LDS buffer, ranked, and written as one 256-B row
//   (+ the centred / normalised xyz of the fused QueryAndGroup path).
#include "common.hip.h"

namespace coda {

constexpr int kCellMax = 32768;       // cells per scene (128 KiB of LDS counters)
constexpr int kGridThreads = 1024;
constexpr int kGridUnroll = 8;         // points per thread whose loads are in flight together
constexpr int kHitCap = 256;          // LDS hit records per wave (4 KiB)
constexpr int kQueryWaves = 4;
constexpr int kGridMinPoints = 1024;  // below this the scan kernel is cheaper than building a grid
constexpr int kGridMaxSample = 128;   // nsample + one 64-wide chunk must fit kHitCap

struct GridHeader {
  float ox, oy, oz, inv_cell;
  int gx, gy, gz, count;
};

inline size_t grid_scene_bytes(int n) {
  return sizeof(GridHeader) + sizeof(int) * (kCellMax + 1) + sizeof(float4) * static_cast<size_t>(n);
}

namespace {

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// Cell coordinate along one axis: floor((v - o) * inv) clamped to [0, g-1].
// Evaluated identically for points and centres; the clamp is monotone, so a hit
// (|dv| < r <= cell/(1+1e-3)) always lies within +-1 cell of the centre's cell.
// (In every distance mode d2 >= fl(dv*dv) for each axis -- the contracted sums only add
// non-negative terms before a monotone rounding -- so d2 < fl(r*r) still bounds |dv| by
// r*(1 + 2^-23), far inside the 1e-3 margin.)
__device__ __forceinline__ int cell_coord(float v, float o, float inv, int g) {
  float u = floorf((v - o) * inv);
  u = fminf(fmaxf(u, 0.0f), static_cast<float>(g - 1));
  return static_cast<int>(u);
}

// LDS counter index with one pad word per 32 counters: thread t scanning its 32 consecutive
// counters then touches banks (t + i) mod 32 -- conflict-free -- instead of a single bank.
__device__ __forceinline__ int cidx(int c) { return c + (c >> 5); }
constexpr int kCntWords = kCellMax + (kCellMax >> 5);

// The three passes over the points (bounding box, histogram, scatter) are latency-bound
// loops (load -> dependent LDS atomic), so each pass handles kGridUnroll points per thread
// per iteration with all their loads issued up front.
__global__ __launch_bounds__(kGridThreads) void grid_build_kernel(const float *__restrict__ xyz,
                                                                  int n, float radius,
                                                                  unsigned char *__restrict__ ws,
                                                                  size_t scene_stride) {
  // one dynamic LDS carve (static LDS is capped at 64 KiB): counters | reductions | header
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *s_cnt = reinterpret_cast<int *>(smem);
  float(*s_red)[kGridThreads / kWave] =
      reinterpret_cast<float(*)[kGridThreads / kWave]>(smem + sizeof(int) * kCntWords);
  int *s_wave_sum = reinterpret_cast<int *>(smem + sizeof(int) * kCntWords + sizeof(float) * 6 * (kGridThreads / kWave));
  GridHeader &s_hdr = *reinterpret_cast<GridHeader *>(smem + sizeof(int) * kCntWords +
                                                      sizeof(float) * 7 * (kGridThreads / kWave));

  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int w = wave_id();
  const float *__restrict__ pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
  unsigned char *base = ws + static_cast<size_t>(blockIdx.x) * scene_stride;
  GridHeader *hdr = reinterpret_cast<GridHeader *>(base);
  int *cell_start = reinterpret_cast<int *>(base + sizeof(GridHeader));
  float4 *records = reinterpret_cast<float4 *>(base + sizeof(GridHeader) + sizeof(int) * (kCellMax + 1));

  // for_each_point(f): f(k, x, y, z) for every point of the scene, kGridUnroll loads in flight
  auto for_each_point = [&](auto &&f) {
    for (int base = 0; base < n; base += kGridThreads * kGridUnroll) {
      float x[kGridUnroll], y[kGridUnroll], z[kGridUnroll];
#pragma unroll
      for (int u = 0; u < kGridUnroll; ++u) {
        const int k = base + u * kGridThreads + tid;
        // out-of-range slots become non-finite and are skipped like NaN/Inf points
        x[u] = k < n ? pts[k * 3 + 0] : INFINITY;
        y[u] = k < n ? pts[k * 3 + 1] : INFINITY;
        z[u] = k < n ? pts[k * 3 + 2] : INFINITY;
      }
#pragma unroll
      for (int u = 0; u < kGridUnroll; ++u) f(base + u * kGridThreads + tid, x[u], y[u], z[u]);
    }
  };

  // ---- 1. bounding box of the finite points
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for_each_point([&](int, float x, float y, float z) {
    if (finite3(x, y, z)) {
      lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
      lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
      lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
    }
  });
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
    if (lane == 0) {
      s_red[a][w] = lo[a];
      s_red[3 + a][w] = hi[a];
    }
  }
  for (int c = tid; c < kCntWords; c += kGridThreads) s_cnt[c] = 0;
  __syncthreads();
  if (tid == 0) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
      mn[a] = INFINITY;
      mx[a] = -INFINITY;
      for (int i = 0; i < kGridThreads / kWave; ++i) {
        mn[a] = fminf(mn[a], s_red[a][i]);
        mx[a] = fmaxf(mx[a], s_red[3 + a][i]);
      }
    }
    GridHeader h;
    h.count = 0;
    if (!(mn[0] <= mx[0])) {  // no finite point at all
      h.ox = h.oy = h.oz = 0.0f;
      h.inv_cell = 0.0f;
      h.gx = h.gy = h.gz = 1;
    } else {
      // cell >= radius * (1 + 1e-3); grow it until the grid fits kCellMax cells
      float cell = fmaxf(radius * 1.001f, 1e-30f);
      int gx, gy, gz;
      for (;;) {
        const float fx = floorf((mx[0] - mn[0]) / cell), fy = floorf((mx[1] - mn[1]) / cell),
                    fz = floorf((mx[2] - mn[2]) / cell);
        if (fx < 16000.0f && fy < 16000.0f && fz < 16000.0f) {
          gx = static_cast<int>(fx) + 1;
          gy = static_cast<int>(fy) + 1;
          gz = static_cast<int>(fz) + 1;
          if (static_cast<long long>(gx) * gy * gz <= kCellMax) break;
        }
        cell *= 1.1f;
      }
      h.ox = mn[0]; h.oy = mn[1]; h.oz = mn[2];
      h.inv_cell = 1.0f / cell;
      h.gx = gx; h.gy = gy; h.gz = gz;
    }
    s_hdr = h;
  }
  __syncthreads();
  const GridHeader h = s_hdr;
  const int ncell = h.gx * h.gy * h.gz;

  // ---- 2. histogram
  for_each_point([&](int, float x, float y, float z) {
    if (finite3(x, y, z)) {
      const int c = (cell_coord(z, h.oz, h.inv_cell, h.gz) * h.gy + cell_coord(y, h.oy, h.inv_cell, h.gy)) * h.gx +
                    cell_coord(x, h.ox, h.inv_cell, h.gx);
      atomicAdd(&s_cnt[cidx(c)], 1);
    }
  });
  __syncthreads();

  // ---- 3. exclusive scan over kCellMax counters (32 consecutive per thread)
  constexpr int kPer = kCellMax / kGridThreads;
  static_assert(kPer == 32, "cidx() padding assumes 32 counters per thread");
  int sum = 0;
#pragma unroll 8
  for (int i = 0; i < kPer; ++i) sum += s_cnt[cidx(tid * kPer + i)];
  int incl = sum;
  for (int off = 1; off < kWave; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == kWave - 1) s_wave_sum[w] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int i = 0; i < w; ++i) wave_off += s_wave_sum[i];
  int run = wave_off + incl - sum;
#pragma unroll 8
  for (int i = 0; i < kPer; ++i) {
    const int c = tid * kPer + i;
    const int cnt = s_cnt[cidx(c)];
    s_cnt[cidx(c)] = run;  // becomes the scatter cursor
    if (c <= ncell) cell_start[c] = run;
    run += cnt;
  }
  if (tid == kGridThreads - 1) {
    if (ncell == kCellMax) cell_start[kCellMax] = run;
    GridHeader out = h;
    out.count = run;
    *hdr = out;
  }
  __syncthreads();

  // ---- 4. scatter (x, y, z, index) records into cell order (order inside a cell is
  //         arbitrary; the query ranks hits by index)
  for_each_point([&](int k, float x, float y, float z) {
    if (finite3(x, y, z)) {
      const int c = (cell_coord(z, h.oz, h.inv_cell, h.gz) * h.gy + cell_coord(y, h.oy, h.inv_cell, h.gy)) * h.gx +
                    cell_coord(x, h.ox, h.inv_cell, h.gx);
      const int pos = atomicAdd(&s_cnt[cidx(c)], 1);
      records[pos] = make_float4(x, y, z, __int_as_float(k));
    }
  });
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
    const unsigned char *__restrict__ ws, size_t scene_stride, int32_t *__restrict__ idx,
    float *__restrict__ grouped, int m, float r2, float inv_radius, int nsample, int normalize,
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

  const unsigned char *base = ws + static_cast<size_t>(bi) * scene_stride;
  const GridHeader h = *reinterpret_cast<const GridHeader *>(base);
  const int *__restrict__ cell_start = reinterpret_cast<const int *>(base + sizeof(GridHeader));
  const float4 *__restrict__ records =
      reinterpret_cast<const float4 *>(base + sizeof(GridHeader) + sizeof(int) * (kCellMax + 1));
  float4 *buf = s_hits[w];

  const float *ctr = new_xyz + (static_cast<size_t>(bi) * m + j) * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int icx = cell_coord(cx, h.ox, h.inv_cell, h.gx);
  const int icy = cell_coord(cy, h.oy, h.inv_cell, h.gy);
  const int icz = cell_coord(cz, h.oz, h.inv_cell, h.gz);

  // lanes 0..8: one (dy, dz) row of up to three x-adjacent (memory-contiguous) cells
  int start = 0, len = 0;
  if (lane < 9 && h.count > 0) {
    const int iy = icy + (lane % 3) - 1, iz = icz + (lane / 3) - 1;
    if (iy >= 0 && iy < h.gy && iz >= 0 && iz < h.gz) {
      const int x0 = max(icx - 1, 0), x1 = min(icx + 1, h.gx - 1);
      const int c0 = (iz * h.gy + iy) * h.gx + x0;
      start = cell_start[c0];
      len = cell_start[c0 + (x1 - x0 + 1)] - start;
    }
  }
  int incl = len;
  for (int off = 1; off < 16; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const int total = __builtin_amdgcn_readlane(incl, 8);

  const uint64_t below = (1ull << lane) - 1ull;
  int nhits = 0;
  for (int t0 = 0; t0 < total; t0 += kWave) {
    const int t = t0 + lane;
    int src = -1;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int r_incl = __builtin_amdgcn_readlane(incl, r);
      const int r_len = __builtin_amdgcn_readlane(len, r);
      const int r_start = __builtin_amdgcn_readlane(start, r);
      if (src < 0 && t < r_incl) src = r_start + (t - (r_incl - r_len));
    }
    const bool valid = t < total;
    float4 p = make_float4(0, 0, 0, 0);
    if (valid) p = records[src];
    const float d2 = sqdist3<DM>(__fsub_rn(cx, p.x), __fsub_rn(cy, p.y), __fsub_rn(cz, p.z));
    const bool hit = valid && d2 < r2;  // same fp32 expression as the scan: ball_query_gpu.cu:34-36
    const uint64_t mask = __ballot(hit);
    const int add = __popcll(mask);
    if (add) {
      if (nhits + add > kHitCap) nhits = rank_and_keep(buf, nhits, nsample, lane);
      if (hit) buf[nhits + __popcll(mask & below)] = p;
      nhits += add;
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
  clear_sticky_error();
  const size_t lds = sizeof(int) * kCntWords + sizeof(float) * 7 * (kGridThreads / kWave) + sizeof(GridHeader);
  auto launch_build = [&](auto kern) -> int {
    static bool lds_set = false;  // once per process: keeps the launch path graph-capturable
    if (!lds_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return static_cast<int>(e);
      lds_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(kGridThreads), lds, s, xyz, n, radius, ws, stride);
    return CODA_OK;
  };
  const int st = launch_build(grid_build_kernel);
  if (st != CODA_OK) return st;
  const float r2 = radius * radius;
  CODA_DISPATCH_DM(distance_mode(),
                   hipLaunchKernelGGL(grid_query_kernel<DM>, dim3(ceil_div(m, kQueryWaves) * b),
                                      dim3(kQueryWaves * kWave), 0, s, new_xyz, xyz, n, ws, stride, idx, grouped, m,
                                      r2, 1.0f / radius, nsample, normalize, b));
  return launch_status();
}

size_t ball_query_grid_workspace(int b, int n, int nsample) {
  if (n < kGridMinPoints || nsample > kGridMaxSample) return 0;
  const size_t stride = (grid_scene_bytes(n) + 255) & ~static_cast<size_t>(255);
  return stride * static_cast<size_t>(b);
}

}  // namespace coda
