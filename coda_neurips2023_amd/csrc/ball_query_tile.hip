// ball_query_tile.hip -- radius search (+ fused xyz grouping) as ONE launch, no workspace, no global atomics.
//
// Replaces, for clouds of >= 1024 points, the two dependent launches of ball_query_grid.hip (a chip-wide cell table
// built in global memory, then one wave per centre chasing cell ranges through three dependent global round trips and
// ranking its hits by index): measured 18 + 22 us at B = 8 with 1.43x the algorithmic HBM traffic (the table).
//
// Here the plane is cut into SUPER-TILES of 4 x 4 lattice cells (cell = 1.001 r, the lattice of ball_query_grid.hip
// folded onto a 32 x 32 torus in x, y; all of z), one workgroup per (scene, super-tile):
//   A  the workgroup scans the scene's M centres and keeps those of its tile, bucketed by SUB-TILE (2 x 2 cells);
//      a tile without centres exits here;
//   B  it streams the scene's cloud ONCE (from the XCD's L2: all tiles of a scene run on one XCD) and keeps the points
//      of its tile + a one-cell halo in LDS -- every point within r of one of its centres is among them -- IN ASCENDING
//      POINT INDEX: each wave takes a quarter of the cloud, compacts 256 points per step with a ballot prefix, and
//      appends to 64-entry LDS blocks it allocates on demand (the logical order = wave 0's blocks, wave 1's, ...);
//   C  per sub-tile, one wave compacts the (still index-ordered) list of LDS slots within that sub-tile + halo;
//   D  the waves take (sub-tile, 8 centres) work items: lane = candidate, centre = scalar, the 64-bit ballot of
//      `d2 < r2` IS the in-index-order hit list (ball_query_gpu.cu:30-43) -- no ranking, early exit at nsample hits;
//      rows are written as full 256-B lines together with the centred / normalised xyz read back from LDS.
// Candidates per centre: the ~4 % of the cloud in a 0.8 m x 0.8 m column instead of all of it (brute force) -- and
// no table in HBM: the kernel reads the cloud and the centres and writes the result, i.e. the algorithmic bytes.
// Exactness: the same fp32 distance expression, strict `<`, first nsample hits in index order, padded with the
// first; bit-identical to the scan kernel.  Degenerate clouds (everything in one tile: the LDS list overflows) fall
// back, per workgroup, to the scan over the whole cloud for that tile's centres.
#include "common.hip.h"

#include <cstdlib>

namespace coda {

namespace {

constexpr int kTqThreads = 256, kTqWaves = kTqThreads / kWave, kTqC = 8;
constexpr int kTorus = 32;                       // lattice cells per axis of the (x, y) torus: 6.4 m at r = 0.2 (larger scenes alias: more candidates, same result)
constexpr int kTileCells = 4;                    // cells per super-tile edge; sub-tiles are 2 x 2 cells
constexpr int kTilesAxis = kTorus / kTileCells;  // 16
constexpr int kTilesScene = kTilesAxis * kTilesAxis;
constexpr int kBlk = 64;                         // LDS block: one candidate per lane
constexpr int kMaxBlocks = 112;

__device__ __forceinline__ int tq_lattice(float v, float inv_cell) {  // = ball_query_grid.hip's lattice()
  int r;
  const float u = __fmul_rn(v, inv_cell);
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(u));
  return r;
}

struct TqLayout {  // byte offsets into dynamic LDS
  int pts, sub, ctr, rows, wlist, order, ocnt, ints, total;
};
__host__ __device__ inline TqLayout tq_layout(int nblk, int cap_s, int m, int nsample) {
  TqLayout l;
  int at = 0;
  l.pts = at; at += 16 * kBlk * nblk;
  l.rows = at; at += 4 * kTqWaves * kTqC * nsample;
  l.ints = at; at += 4 * 32;
  l.sub = at; at += 2 * 4 * cap_s;
  l.ctr = at; at += 2 * ((m + 7) & ~7);
  l.order = at; at += 2 * ((nblk + 7) & ~7);
  l.wlist = at; at += kTqWaves * ((nblk + 15) & ~15);
  l.ocnt = at; at += (nblk + 15) & ~15;
  l.total = (at + 15) & ~15;
  return l;
}
// indices into the small-int area
enum { I_CCNT = 0, I_CBASE = 4, I_CFILL = 9, I_NBLK = 13, I_WN = 14, I_WFILL = 18, I_SCNT = 22, I_OVF = 26, I_WORK = 27 };

struct Quad4 {
  float v[12];
};
__device__ __forceinline__ Quad4 tq_load_quad(const float *__restrict__ pts, int g, int n, bool vec) {
  Quad4 q;
  const int k0 = 4 * g;
  if (vec && k0 + 3 < n) {
    const float4 a = *reinterpret_cast<const float4 *>(pts + k0 * 3), b = *reinterpret_cast<const float4 *>(pts + k0 * 3 + 4),
                 c = *reinterpret_cast<const float4 *>(pts + k0 * 3 + 8);
    q.v[0] = a.x; q.v[1] = a.y; q.v[2] = a.z; q.v[3] = a.w; q.v[4] = b.x; q.v[5] = b.y; q.v[6] = b.z; q.v[7] = b.w;
    q.v[8] = c.x; q.v[9] = c.y; q.v[10] = c.z; q.v[11] = c.w;
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) q.v[i] = (k0 * 3 + i < n * 3) ? pts[k0 * 3 + i] : 0.0f;
  }
  return q;
}

template <int DM>
__global__ __launch_bounds__(kTqThreads) void ball_query_tile_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int32_t *__restrict__ idx, float *__restrict__ grouped,
    int n, int m, float r2, float inv_radius, float inv_cell, int nsample, int normalize, int nscenes, int nblk, int cap_s,
    int vec, int stop_after) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const TqLayout L = tq_layout(nblk, cap_s, m, nsample);
  float4 *s_pts = reinterpret_cast<float4 *>(smem + L.pts);
  unsigned short *s_sub = reinterpret_cast<unsigned short *>(smem + L.sub);    // [4][cap_s] LDS slots, index-ordered
  unsigned short *s_ctr = reinterpret_cast<unsigned short *>(smem + L.ctr);    // centre indices bucketed by sub-tile
  int32_t *s_rows = reinterpret_cast<int32_t *>(smem + L.rows);               // [waves][C][nsample]
  unsigned char *s_wlist = smem + L.wlist;                                    // [waves][nblk16] block ids in order
  unsigned short *s_order = reinterpret_cast<unsigned short *>(smem + L.order);  // logical block order
  unsigned char *s_ocnt = smem + L.ocnt;                                      // entries of each ordered block
  int *s_i = reinterpret_cast<int *>(smem + L.ints);
  const int wl_stride = (nblk + 15) & ~15;

  const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
  // XCD-aware: workgroup g runs on XCD g % 8 (observed dispatch order), so scene = g % B keeps all tiles of a scene on
  // one XCD's L2 -- the cloud comes from HBM once.  Only speed depends on the placement.
  const int scene = blockIdx.x % nscenes, tile = blockIdx.x / nscenes;
  const int tx = tile % kTilesAxis, ty = tile / kTilesAxis;
  const float *__restrict__ pts = xyz + static_cast<size_t>(scene) * n * 3;
  const float *__restrict__ ctr = new_xyz + static_cast<size_t>(scene) * m * 3;

  if (tid < 32) s_i[tid] = 0;
  __syncthreads();

  // ---- A: the centres of this tile, by sub-tile ------------------------------------------------------------------
  auto code_of = [&](float x, float y) {  // sub-tile 0..3 of a centre if it lies in this tile, else -1
    const int icx = tq_lattice(x, inv_cell), icy = tq_lattice(y, inv_cell);
    const bool mine = ((icx >> 2) & (kTilesAxis - 1)) == tx && ((icy >> 2) & (kTilesAxis - 1)) == ty;
    return mine ? (((icy >> 1) & 1) * 2 + ((icx >> 1) & 1)) : -1;
  };
  constexpr int kCpt = 8;  // centres per thread and batch: their loads are in flight together
  for (int jb = 0; jb < m; jb += kTqThreads * kCpt) {
    float x[kCpt], y[kCpt];
#pragma unroll
    for (int i = 0; i < kCpt; ++i) {
      const int j = min(jb + i * kTqThreads + tid, m - 1);
      x[i] = ctr[j * 3 + 0];
      y[i] = ctr[j * 3 + 1];
    }
#pragma unroll
    for (int i = 0; i < kCpt; ++i) {
      const int s = jb + i * kTqThreads + tid < m ? code_of(x[i], y[i]) : -1;
      if (s >= 0) atomicAdd(&s_i[I_CCNT + s], 1);
    }
  }
  __syncthreads();
  const int c0 = s_i[I_CCNT], c1 = s_i[I_CCNT + 1], c2 = s_i[I_CCNT + 2], c3 = s_i[I_CCNT + 3];
  const int ctotal = c0 + c1 + c2 + c3;
  if (ctotal == 0) return;  // block-uniform: nothing to answer here
  if (tid == 0) {
    s_i[I_CBASE] = 0; s_i[I_CBASE + 1] = c0; s_i[I_CBASE + 2] = c0 + c1; s_i[I_CBASE + 3] = c0 + c1 + c2;
    s_i[I_CBASE + 4] = ctotal;
  }
  __syncthreads();
  for (int jb = 0; jb < m; jb += kTqThreads * kCpt) {
    float x[kCpt], y[kCpt];
#pragma unroll
    for (int i = 0; i < kCpt; ++i) {
      const int j = min(jb + i * kTqThreads + tid, m - 1);
      x[i] = ctr[j * 3 + 0];
      y[i] = ctr[j * 3 + 1];
    }
#pragma unroll
    for (int i = 0; i < kCpt; ++i) {
      const int j = jb + i * kTqThreads + tid;
      const int s = j < m ? code_of(x[i], y[i]) : -1;
      if (s >= 0) s_ctr[s_i[I_CBASE + s] + atomicAdd(&s_i[I_CFILL + s], 1)] = static_cast<unsigned short>(j);
    }
  }

  if (stop_after == 1) return;
  // ---- B: the points of tile + halo into LDS, in ascending point index -----------------------------------------
  {
    const int ox = tx * kTileCells - 1, oy = ty * kTileCells - 1;  // lattice origin of the haloed tile (6 x 6 cells)
    const int ngroups = (n + 3) / 4;
    const int per = ((ngroups + kTqWaves - 1) / kTqWaves + kWave - 1) / kWave * kWave;
    const int gbeg = w * per, gend = min(ngroups, gbeg + per);
    int cur = 0, fill = kBlk, wn = 0;  // fill = kBlk: the first member allocates a block
    bool dead = false;                 // LDS exhausted: the whole workgroup falls back to the global scan
    constexpr int kUn = 4;  // steps (of 256 points per wave) whose loads are issued together: 12 x 16 B per lane in flight
    for (int gb = gbeg; gb < gend && !dead; gb += kWave * kUn) {
      Quad4 qs[kUn];
#pragma unroll
      for (int un = 0; un < kUn; ++un) {
        const int gq = gb + un * kWave + lane;
        qs[un] = tq_load_quad(pts, gq < gend ? gq : gbeg, n, vec != 0);
      }
#pragma unroll
      for (int un = 0; un < kUn; ++un) {
        const int g0 = gb + un * kWave;
        if (g0 >= gend || dead) break;  // wave-uniform
        const Quad4 &q = qs[un];
        const int g = g0 + lane;
        const bool qv = g < gend;
        int code[4], cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int u = (tq_lattice(q.v[3 * j], inv_cell) - ox) & (kTorus - 1);
          const int v = (tq_lattice(q.v[3 * j + 1], inv_cell) - oy) & (kTorus - 1);
          const bool in = qv && 4 * g + j < n && u < kTileCells + 2 && v < kTileCells + 2;
          // sub-tile + halo = 4 x 4 cells: x in [0,4) for sx = 0, [2,6) for sx = 1; same in y
          const int bx = (u < 4 ? 1 : 0) | (u >= 2 ? 2 : 0), by = (v < 4 ? 1 : 0) | (v >= 2 ? 2 : 0);
          const int sm = ((bx & 1) && (by & 1) ? 1 : 0) | ((bx & 2) && (by & 1) ? 2 : 0) | ((bx & 1) && (by & 2) ? 4 : 0) |
                         ((bx & 2) && (by & 2) ? 8 : 0);
          code[j] = in ? (sm | 16) : 0;
          cnt += in ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
          const int up = __shfl_up(incl, off);
          if (lane >= off) incl += up;
        }
        const int tot = __builtin_amdgcn_readlane(incl, kWave - 1);
        if (tot == 0) continue;
        const int nnew = (fill + tot - 1) >> 6;  // blocks needed beyond the current one
        int nb = 0;
        if (nnew > 0) {
          if (lane == 0) nb = atomicAdd(&s_i[I_NBLK], nnew);
          nb = __builtin_amdgcn_readfirstlane(nb);
          if (nb + nnew > nblk) {
            if (lane == 0) s_i[I_OVF] = 1;
            dead = true;
            break;
          }
          if (lane < nnew) s_wlist[w * wl_stride + wn + lane] = static_cast<unsigned char>(nb + lane);
          wn += nnew;
        }
        int pos = fill + incl - cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (code[j]) {
            const int bsel = pos >> 6;
            const int slot = (bsel == 0 ? cur : nb + bsel - 1) * kBlk + (pos & 63);
            s_pts[slot] = make_float4(q.v[3 * j], q.v[3 * j + 1], q.v[3 * j + 2],
                                      __int_as_float((4 * g + j) | ((code[j] & 15) << 24)));
            ++pos;
          }
        }
        if (nnew > 0) cur = nb + nnew - 1;
        fill = ((fill + tot - 1) & 63) + 1;
      }
    }
    if (lane == 0) {
      s_i[I_WN + w] = wn;
      s_i[I_WFILL + w] = fill;
    }
  }
  __syncthreads();
  if (stop_after == 2) return;
  const bool overflow = s_i[I_OVF] != 0;
  const int nbt = overflow ? 0 : min(s_i[I_NBLK], nblk);
  if (!overflow && tid < kTqWaves) {  // the logical order of the blocks: wave 0's, wave 1's, ...
    int at = 0;
    for (int q = 0; q < tid; ++q) at += s_i[I_WN + q];
    const int wn = s_i[I_WN + tid];
    for (int i = 0; i < wn; ++i) {
      s_order[at + i] = s_wlist[tid * wl_stride + i];
      s_ocnt[at + i] = static_cast<unsigned char>(i == wn - 1 ? s_i[I_WFILL + tid] : kBlk);
    }
  }
  __syncthreads();

  // ---- C: per sub-tile, the index-ordered list of its candidates' LDS slots (wave s builds list s) -----------------
  const uint64_t below = (1ull << lane) - 1ull;
  if (!overflow) {
    int cs = 0;
    unsigned short *mine = s_sub + w * cap_s;
    for (int k = 0; k < nbt; ++k) {
      const int slot = s_order[k] * kBlk + lane;
      const bool bit = lane < s_ocnt[k] && ((__float_as_int(s_pts[slot].w) >> (24 + w)) & 1);
      const uint64_t mask = __ballot(bit);
      if (bit) {
        const int pos = cs + __popcll(mask & below);
        if (pos < cap_s) mine[pos] = static_cast<unsigned short>(slot);
      }
      cs += __popcll(mask);
    }
    if (lane == 0) s_i[I_SCNT + w] = cs;  // > cap_s: that sub-tile's centres walk the whole tile list instead
  }
  __syncthreads();

  if (stop_after == 3) return;
  // ---- D: (sub-tile, 8 centres) work items --------------------------------------------------------------------
  int ng[4], ntotal = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ng[s] = (s_i[I_CCNT + s] + kTqC - 1) / kTqC;
    ntotal += ng[s];
  }
  int32_t *rows = s_rows + static_cast<size_t>(w) * kTqC * nsample;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(&s_i[I_WORK], 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= ntotal) break;
    int s = 0;
    while (item >= ng[s]) {
      item -= ng[s];
      ++s;
    }
    const int cbeg = s_i[I_CBASE + s] + item * kTqC;
    const int cnum = min(kTqC, s_i[I_CBASE + s + 1] - cbeg);
    const int scnt = s_i[I_SCNT + s];
    // candidate source: 0 = the sub-tile's slot list, 1 = every block of the tile, 2 = the whole cloud (global)
    const int mode = overflow ? 2 : (scnt > cap_s ? 1 : 0);
    const int niter = mode == 0 ? (scnt + kWave - 1) / kWave : (mode == 1 ? nbt : (n + kWave - 1) / kWave);
    const unsigned short *slist = s_sub + s * cap_s;

    int cj[kTqC];
    float cx[kTqC], cy[kTqC], cz[kTqC];
    int cnt[kTqC], first[kTqC];
#pragma unroll
    for (int c = 0; c < kTqC; ++c) {
      cj[c] = s_ctr[cbeg + min(c, cnum - 1)];
      cx[c] = ctr[cj[c] * 3 + 0];
      cy[c] = ctr[cj[c] * 3 + 1];
      cz[c] = ctr[cj[c] * 3 + 2];
      cnt[c] = c < cnum ? 0 : nsample;  // tail slots start "full"
      first[c] = 0;
    }
    for (int it = 0; it < niter; ++it) {
      bool valid;
      int key;
      float x, y, z;
      if (mode == 0) {
        const int i = it * kWave + lane;
        valid = i < scnt;
        key = slist[valid ? i : 0];
      } else if (mode == 1) {
        valid = lane < s_ocnt[it];
        key = s_order[it] * kBlk + lane;
      } else {
        key = it * kWave + lane;
        valid = key < n;
        if (!valid) key = n - 1;
      }
      if (mode == 2) {
        x = pts[key * 3 + 0]; y = pts[key * 3 + 1]; z = pts[key * 3 + 2];
      } else {
        const float4 p = s_pts[key];
        x = p.x; y = p.y; z = p.z;
      }
      bool all_full = true;
#pragma unroll
      for (int c = 0; c < kTqC; ++c) {
        if (cnt[c] < nsample) {  // ball_query_gpu.cu:30 `cnt < nsample`
          const float d2 = sqdist3<DM>(__fsub_rn(cx[c], x), __fsub_rn(cy[c], y), __fsub_rn(cz[c], z));
          const bool hit = valid && d2 < r2;  // :36 strict
          const uint64_t mask = __ballot(hit);
          if (mask) {
            const int pos = cnt[c] + __popcll(mask & below);
            if (hit && pos < nsample) rows[c * nsample + pos] = key;  // :42
            if (cnt[c] == 0) first[c] = __shfl(key, __ffsll(static_cast<unsigned long long>(mask)) - 1);
            cnt[c] += __popcll(mask);
          }
          all_full = all_full && (cnt[c] >= nsample);
        }
      }
      if (all_full) break;
    }
    __builtin_amdgcn_wave_barrier();  // rows[] written by other lanes of this wave

#pragma unroll
    for (int c = 0; c < kTqC; ++c) {
      if (c < cnum) {
        const int j = cj[c];
        const int filled = min(cnt[c], nsample);
        const size_t row_off = (static_cast<size_t>(scene) * m + j) * nsample;
        const size_t plane = static_cast<size_t>(m) * nsample;
        for (int sidx = lane; sidx < nsample; sidx += kWave) {
          // pad with the first hit (:37-41); an empty ball keeps index 0 (zero-filled output) = point 0 of the scene
          float px, py, pz;
          int v;
          if (filled == 0) {
            v = 0;
            px = pts[0]; py = pts[1]; pz = pts[2];
          } else {
            const int key = sidx < filled ? rows[c * nsample + sidx] : first[c];
            if (mode == 2) {
              v = key;
              px = pts[key * 3 + 0]; py = pts[key * 3 + 1]; pz = pts[key * 3 + 2];
            } else {
              const float4 p = s_pts[key];
              v = __float_as_int(p.w) & 0xffffff;
              px = p.x; py = p.y; pz = p.z;
            }
          }
          idx[row_off + sidx] = v;
          if (grouped) {
            // grouped_xyz -= new_xyz (pointnet2_utils.py:347); /= radius (:348-349: a multiply by the fp32 reciprocal)
            float gx = __fsub_rn(px, cx[c]), gy = __fsub_rn(py, cy[c]), gz = __fsub_rn(pz, cz[c]);
            if (normalize & 1) {
              gx = __fmul_rn(gx, inv_radius); gy = __fmul_rn(gy, inv_radius); gz = __fmul_rn(gz, inv_radius);
            }
            if (normalize & 2) {  // channels-last (B,M,S,3)
              float *gp = grouped + (row_off + sidx) * 3;
              gp[0] = gx; gp[1] = gy; gp[2] = gz;
            } else {
              float *gp = grouped + static_cast<size_t>(scene) * 3 * plane + static_cast<size_t>(j) * nsample + sidx;
              gp[0] = gx;
              gp[plane] = gy;
              gp[2 * plane] = gz;
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // rows[] are reused by the next item
  }
}

}  // namespace

bool ball_query_tile_applies(int n, int m, int nsample) {
  return n >= 1024 && n < (1 << 24) && m <= 65535 && nsample <= 256;
}

int ball_query_tile(const float *new_xyz, const float *xyz, int32_t *idx, float *grouped, int b, int n, int m, float radius,
                    int nsample, int normalize, hipStream_t s) {
  const float cell = fmaxf(radius * 1.001f, 1e-30f);
  const float inv_cell = 1.0f / cell;
  const float r2 = radius * radius;  // ball_query_gpu.cu:25 (fp32 product)
  // LDS blocks for the tile's points: a sixth of the cloud (a haloed tile of an indoor scene holds ~7 %), within 160 KiB
  int nblk = (n / 6 + kBlk - 1) / kBlk;
  nblk = nblk < 16 ? 16 : (nblk > kMaxBlocks ? kMaxBlocks : nblk);
  int cap_s = nblk * kBlk / 2;
  while (nblk > 16 && tq_layout(nblk, cap_s, m, nsample).total > 156 * 1024) {
    nblk -= 8;
    cap_s = nblk * kBlk / 2;
  }
  const size_t lds = tq_layout(nblk, cap_s, m, nsample).total;
  if (lds > 160 * 1024) return CODA_ENOSPC;
  const int vec = (n % 4 == 0 && (reinterpret_cast<uintptr_t>(xyz) & 15) == 0) ? 1 : 0;
  static const int stop_after = [] { const char *e = getenv("CODA_BQ_TILE_STOP"); return e ? atoi(e) : 0; }();  // dev: phase timing
  clear_sticky_error();
  int st = CODA_OK;
  CODA_DISPATCH_DM(distance_mode(), {
    auto kern = ball_query_tile_kernel<DM>;
    st = raise_dynamic_lds(kern, lds);
    if (st == CODA_OK)
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(b) * kTilesScene), dim3(kTqThreads), lds, s, new_xyz, xyz, idx,
                         grouped, n, m, r2, 1.0f / radius, inv_cell, nsample, normalize, b, nblk, cap_s, vec, stop_after);
  });
  return st != CODA_OK ? st : launch_status();
}

}  // namespace coda
