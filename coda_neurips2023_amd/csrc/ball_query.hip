// ball_query.hip -- radius neighbour search (+ fused xyz grouping) for gfx950.
//
// Replaces query_ball_point_kernel (third_party_pointnet2/pointnet2/_ext_src/
// src/ball_query_gpu.cu:12-57: one 512-thread block per scene, every thread
// scanning all N points serially for its centres) and, in the fused entry
// point, the transpose + group_points + centre + 1/radius passes of
// QueryAndGroup.forward (pointnet2_utils.py:331-349).
//
// Scan kernel (this file, "brute force" but wave64-shaped):
//  * lane = point, centre = wave-uniform.  A wave loads 64 consecutive points
//    (coalesced 768 B) and tests them against C centres held in SGPRs; the
//    64-bit ballot of `d2 < r2` IS the in-index-order hit list of that chunk,
//    so "first nsample hits in ascending index order" (ball_query_gpu.cu:30-43)
//    falls out of mbcnt prefix counts with no sorting and no atomics.
//  * per-centre early exit once nsample hits are found (:30), wave exit when
//    all C centres are full.
//  * hit rows are staged in LDS and written as full 256-B rows; the fused
//    variant gathers xyz[idx] while the row is still on chip and emits the
//    centred / normalised (B,3,M,S) tensor directly.
#include "common.hip.h"

#include <atomic>
#include <cstdlib>

namespace coda {

// ball_query_grid.hip
int ball_query_grid(const float *new_xyz, const float *xyz, int32_t *idx, float *grouped, int b, int n,
                    int m, float radius, int nsample, int normalize, void *workspace, hipStream_t s);
size_t ball_query_grid_workspace(int b, int n, int nsample);
namespace {
int ball_query_route() {
  const int r = call_options().bq_route;  // this call's option (coda_ball_query_opt_f32), else CODA_BQ = auto | grid | scan
  if (r >= 1 && r <= 2) return r;
  static const int dflt = [] {
    const char *e = getenv("CODA_BQ");
    return !e ? 0 : (e[0] == 'g' ? 1 : (e[0] == 's' ? 2 : 0));
  }();
  return dflt;
}

constexpr int kBqWaves = 4;  // waves per workgroup

template <int C, int DM>
__global__ __launch_bounds__(kBqWaves * kWave) void ball_query_scan_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int32_t *__restrict__ idx,
    float *__restrict__ grouped, int n, int m, float r2, float inv_radius, int nsample,
    int normalize, int nscenes) {
  extern __shared__ __attribute__((aligned(16))) int32_t s_rows[];  // [waves][C][nsample]

  const int w = wave_id();
  const int lane = lane_id();
  // scene = workgroup id % B keeps a scene on one XCD's L2 (see ball_query_grid.hip)
  const int bi = blockIdx.x % nscenes;
  const int j0 = ((blockIdx.x / nscenes) * kBqWaves + w) * C;
  if (j0 >= m) return;  // wave-uniform; no workgroup barrier below

  const float *__restrict__ pts = xyz + static_cast<size_t>(bi) * n * 3;
  const float *__restrict__ ctr = new_xyz + static_cast<size_t>(bi) * m * 3;
  int32_t *rows = s_rows + static_cast<size_t>(w) * C * nsample;

  float cx[C], cy[C], cz[C];
  int cnt[C], first[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = min(j0 + c, m - 1);
    cx[c] = ctr[j * 3 + 0];
    cy[c] = ctr[j * 3 + 1];
    cz[c] = ctr[j * 3 + 2];
    cnt[c] = (j0 + c < m) ? 0 : nsample;  // tail slots start "full"
    first[c] = 0;
  }

  const uint64_t below = (1ull << lane) - 1ull;
  for (int k0 = 0; k0 < n; k0 += kWave) {
    const int k = k0 + lane;
    const bool valid = k < n;
    const int kk = valid ? k : n - 1;
    const float x = pts[kk * 3 + 0];
    const float y = pts[kk * 3 + 1];
    const float z = pts[kk * 3 + 2];
    bool all_full = true;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (cnt[c] < nsample) {  // ball_query_gpu.cu:30 `cnt < nsample`
        const float d2 = sqdist3<DM>(__fsub_rn(cx[c], x), __fsub_rn(cy[c], y), __fsub_rn(cz[c], z));
        const bool hit = valid && d2 < r2;  // :36 strict
        const uint64_t mask = __ballot(hit);
        if (mask) {
          const int pos = cnt[c] + __popcll(mask & below);
          if (hit && pos < nsample) rows[c * nsample + pos] = k;  // :42
          if (cnt[c] == 0) first[c] = k0 + __ffsll(static_cast<unsigned long long>(mask)) - 1;
          cnt[c] += __popcll(mask);
        }
        all_full = all_full && (cnt[c] >= nsample);
      }
    }
    if (all_full) break;
  }
  __builtin_amdgcn_wave_barrier();  // rows[] written by other lanes of this wave

#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = j0 + c;
    if (j < m) {
      const int filled = min(cnt[c], nsample);
      const int fillv = first[c];  // :37-41 first hit; 0 when empty (zero-filled output)
      const size_t row_off = (static_cast<size_t>(bi) * m + j) * nsample;
      for (int s = lane; s < nsample; s += kWave) {
        const int v = s < filled ? rows[c * nsample + s] : fillv;
        idx[row_off + s] = v;
        if (grouped) {
          // grouped_xyz -= new_xyz (pointnet2_utils.py:347); /= radius (:348-349),
          // which torch evaluates on the GPU as a multiply by the fp32 reciprocal.
          float gx = __fsub_rn(pts[v * 3 + 0], cx[c]);
          float gy = __fsub_rn(pts[v * 3 + 1], cy[c]);
          float gz = __fsub_rn(pts[v * 3 + 2], cz[c]);
          if (normalize & 1) {
            gx = __fmul_rn(gx, inv_radius);
            gy = __fmul_rn(gy, inv_radius);
            gz = __fmul_rn(gz, inv_radius);
          }
          if (normalize & 2) {  // channels-last (B,M,S,3): feeds the fused shared MLP
            float *g = grouped + ((static_cast<size_t>(bi) * m + j) * nsample + s) * 3;
            g[0] = gx; g[1] = gy; g[2] = gz;
          } else {              // (B,3,M,S): the reference layout
            const size_t plane = static_cast<size_t>(m) * nsample;
            float *g = grouped + static_cast<size_t>(bi) * 3 * plane + static_cast<size_t>(j) * nsample + s;
            g[0] = gx;
            g[plane] = gy;
            g[2 * plane] = gz;
          }
        }
      }
    }
  }
}

template <int C>
int launch_scan(const float *new_xyz, const float *xyz, int32_t *idx, float *grouped, int b, int n,
                int m, float radius, int nsample, int normalize, hipStream_t s) {
  const size_t lds = sizeof(int32_t) * kBqWaves * C * static_cast<size_t>(nsample);
  const float r2 = radius * radius;  // ball_query_gpu.cu:25 (fp32 product)
  const float inv_radius = 1.0f / radius;
  dim3 grid(ceil_div(m, kBqWaves * C) * b);
  clear_sticky_error();
  int st = CODA_OK;
  CODA_DISPATCH_DM(distance_mode(), {
    auto kern = ball_query_scan_kernel<C, DM>;
    st = raise_dynamic_lds(kern, lds);
    if (st == CODA_OK)
      hipLaunchKernelGGL(kern, grid, dim3(kBqWaves * kWave), lds, s, new_xyz, xyz, idx, grouped, n, m, r2,
                         inv_radius, nsample, normalize, b);
  });
  return st != CODA_OK ? st : launch_status();
}

int ball_query_dispatch(const float *new_xyz, const float *xyz, int32_t *idx, float *grouped, int b,
                        int n, int m, float radius, int nsample, int normalize, void *workspace,
                        size_t workspace_bytes, hipStream_t s) {
  // 0 auto (grid when the caller provided its workspace, else scan) | 1 grid | 2 scan
  const int route = ball_query_route();
  // cell-binned search when the caller provided the workspace it was told to provide
  const size_t need = ball_query_grid_workspace(b, n, nsample);
  if (route != 2 && workspace && need > 0 && workspace_bytes >= need && radius > 0.0f)
    return ball_query_grid(new_xyz, xyz, idx, grouped, b, n, m, radius, nsample, normalize, workspace, s);
  // LDS rows: waves * C * nsample * 4 B must fit the 160 KiB CU.
  const size_t per_centre = sizeof(int32_t) * kBqWaves * static_cast<size_t>(nsample);
  if (per_centre * 8 <= 64 * 1024)
    return launch_scan<8>(new_xyz, xyz, idx, grouped, b, n, m, radius, nsample, normalize, s);
  if (per_centre * 2 <= 64 * 1024)
    return launch_scan<2>(new_xyz, xyz, idx, grouped, b, n, m, radius, nsample, normalize, s);
  if (per_centre <= 160 * 1024)
    return launch_scan<1>(new_xyz, xyz, idx, grouped, b, n, m, radius, nsample, normalize, s);
  return CODA_EINVAL;  // nsample > 10240
}

}  // namespace
}  // namespace coda


CODA_API size_t coda_ball_query_workspace_bytes(int b, int n, int m, int nsample) {
  (void)m;
  if (b <= 0 || n <= 0 || nsample <= 0) return 0;
  return coda::ball_query_grid_workspace(b, n, nsample);
}

CODA_API int coda_ball_query_f32(const float *new_xyz, const float *xyz, int32_t *idx, int b, int n,
                                 int m, float radius, int nsample, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample <= 0) return CODA_EINVAL;
  if (b == 0 || m == 0) return CODA_OK;
  if (!new_xyz || !xyz || !idx) return CODA_EINVAL;
  return coda::ball_query_dispatch(new_xyz, xyz, idx, nullptr, b, n, m, radius, nsample, 0, workspace,
                                   workspace_bytes, static_cast<hipStream_t>(stream));
}

CODA_API int coda_query_and_group_xyz_f32(const float *new_xyz, const float *xyz, int32_t *idx,
                                          float *grouped_xyz, int b, int n, int m, float radius,
                                          int nsample, int normalize, void *workspace,
                                          size_t workspace_bytes, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample <= 0) return CODA_EINVAL;
  if (b == 0 || m == 0) return CODA_OK;
  if (!new_xyz || !xyz || !idx || !grouped_xyz) return CODA_EINVAL;
  return coda::ball_query_dispatch(new_xyz, xyz, idx, grouped_xyz, b, n, m, radius, nsample,
                                   normalize, workspace, workspace_bytes,
                                   static_cast<hipStream_t>(stream));
}

CODA_API int coda_ball_query_opt_f32(const float *new_xyz, const float *xyz, int32_t *idx, int b, int n, int m,
                                     float radius, int nsample, void *workspace, size_t workspace_bytes,
                                     int distance_mode, int route, void *stream) {
  if (distance_mode < -1 || distance_mode >= coda::kDistanceModes || route < 0 || route > 2) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.distance_mode = distance_mode;
  o.bq_route = route;
  coda::ScopedCallOptions scope(o);
  return coda_ball_query_f32(new_xyz, xyz, idx, b, n, m, radius, nsample, workspace, workspace_bytes, stream);
}

CODA_API int coda_query_and_group_xyz_opt_f32(const float *new_xyz, const float *xyz, int32_t *idx, float *grouped_xyz,
                                              int b, int n, int m, float radius, int nsample, int normalize,
                                              void *workspace, size_t workspace_bytes, int distance_mode, int route,
                                              void *stream) {
  if (distance_mode < -1 || distance_mode >= coda::kDistanceModes || route < 0 || route > 2) return CODA_EINVAL;
  coda::CallOptions o = coda::call_options();
  o.distance_mode = distance_mode;
  o.bq_route = route;
  coda::ScopedCallOptions scope(o);
  return coda_query_and_group_xyz_f32(new_xyz, xyz, idx, grouped_xyz, b, n, m, radius, nsample, normalize, workspace,
                                      workspace_bytes, stream);
}
