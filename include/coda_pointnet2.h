/*
 * coda_pointnet2.h -- C ABI of libcoda_hip.so, the MI355X (gfx950) point-cloud
 * operators of the CoDA / 3DETR set-abstraction path.
 *
 * Each entry point replaces one function of the reference's pybind11 module
 * `pointnet2._ext` (third_party_pointnet2/pointnet2/_ext_src/src/bindings.cpp:9-22);
 * the per-function comments cite the reference C++ wrapper and CUDA kernel.
 *
 * Conventions
 *  - Plain device pointers + sizes, no torch types.  All tensors are dense,
 *    row-major ("contiguous"), float32 / int32, resident in HBM of the current
 *    device.  Inputs are borrowed and never written.
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *    on that stream (no allocation, no synchronisation, graph-capture safe).
 *  - Outputs are fully written by the call (the reference relies on
 *    torch::zeros pre-fill; here the kernels write the fill values themselves),
 *    so callers may pass uninitialised memory.
 *  - Return value: 0 on success; a positive hipError_t if the launch failed;
 *    CODA_EINVAL (-1) for invalid arguments, CODA_ENOSPC (-2) if `workspace`
 *    is too small.  (The reference prints and exit(-1)s on launch failure,
 *    include/cuda_utils.h:32-41; the binding raises instead.)
 *  - Re-entrant and thread-safe; the only process-wide state is the distance
 *    arithmetic mode below (an atomic int read once per call).
 *
 * Arithmetic contract (parity): see "distance arithmetic mode" below.
 */
#ifndef CODA_POINTNET2_H
#define CODA_POINTNET2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CODA_OK 0
#define CODA_EINVAL (-1)
#define CODA_ENOSPC (-2)
#define CODA_ELOST (-3) /* an earlier two-workgroup sampling launch lost its partner workgroup: see below */

/* Library identification: returns "coda_hip gfx950 <abi-version>". */
const char *coda_version(void);

/* ---- distance arithmetic mode ---------------------------------------------------
 * The reference's kernels evaluate  dx*dx + dy*dy + dz*dz  (sampling_gpu.cu:103-107,
 * ball_query_gpu.cu:34-35, interpolate_gpu.cu:36) and  p1*w1 + p2*w2 + p3*w3
 * (interpolate_gpu.cu:101-102) in fp32 inside a binary that nvcc builds with its
 * default -fmad=true (third_party_pointnet2/pointnet2/setup.py:26-28 passes no -fmad
 * flag), i.e. WITH fused-multiply-add contraction.  The contraction nvcc picks cannot
 * be read off the sources, so all three candidates are implemented bit-exactly and the
 * mode is an ARGUMENT of the *_opt entry points below (every distance / interpolation
 * kernel of this library has one; FPS indices, ball_query rows and three_nn change with
 * it only where two candidates are within one rounding of each other):
 *   0  no contraction           (a*a' + b*b') + c*c'           one rounding per operation
 *   1  fma(c,c', fma(a,a', b*b'))   DEFAULT: the first product of the inner sum is
 *                                   contracted, the LLVM / NVVM combiner order
 *   2  fma(c,c', fma(b,b', a*a'))
 *  -1  the library default: environment variable CODA_DISTANCE_MODE (0|1|2) read once
 *      at first use, else 1 -- what the entry points without the argument use.
 * The library has NO mutable process-wide state (the reference's ops have none,
 * ball_query.cpp:11-35): two callers with different modes -- or different threads --
 * never interact.  The CPU oracle (oracle/pointnet2_oracle.c, `fma_mode`) has the same
 * three modes; parity tests run all three.  coda_get_distance_mode() = the default.  */
int coda_get_distance_mode(void);

/* ---- furthest_point_sampling ------------------------------------------------
 * Replaces furthest_point_sampling(points (B,N,3) f32, nsamples) -> (B,m) i32
 *   wrapper  src/sampling.cpp:67-88, kernel src/sampling_gpu.cu:72-232.
 * Semantics kept bit-exactly (given the arithmetic contract):
 *   start at index 0; points with x*x+y*y+z*z <= 1e-3 (double literal) never
 *   take part; running min distance starts at 1e10; arg-max ties resolve like
 *   the reference's 2^k-thread strided scan + LDS tree:  smallest
 *   bit-reversed (k mod T) first, then smallest k, with
 *   T = min(512, 2^floor(log2 N))  (include/cuda_utils.h:17-21).
 * Kernels by size (m >= 128 samples): 4096 <= N <= 20480: one workgroup per
 * scene, the Morton-sorted cloud in its registers, only the buckets a new sample
 * can reach are updated; 20480 < N <= 40960: TWO workgroups per scene, half of
 * the buckets each, exchanging their candidate every round through a mailbox in
 * `workspace` (one relaxed 64-bit atomic each way).  LAUNCH PRECONDITION of the
 * pair, kept by the library itself: both workgroups of a scene must be resident
 * together, so a call is issued as launches of at most 64 scenes (128 workgroups,
 * half of the 256 CUs) on the caller's stream -- any B is accepted.  What the
 * library cannot see is OTHER work of the caller that fills the device with
 * workgroups that never retire; a wait of ~2^22 polls without an answer is
 * abandoned rather than hanging the device -- LOUDLY: see
 * coda_fps_lost_partner_events below); otherwise the running distances live
 * in registers (N <= 24576), LDS (N <= ~40000) or `workspace` (B*N floats, the
 * reference's `tmp` tensor, sampling.cpp:75-77).  coda_..._workspace_bytes()
 * gives the size the chosen kernel needs (Morton records, mailboxes, distances;
 * 256-byte aligned); 0: NULL/0 may be passed.  Without a sufficient workspace
 * the call falls back to the kernels that need none.
 * coda_furthest_point_sampling_opt_f32: the same with per-call options --
 * distance_mode as above, waves (0 default | 8 | 16) = waves per workgroup of the
 * bucketed kernels (same indices either way; default: CODA_FPS_WAVES, else 16 with
 * one workgroup per scene, 8 with two).                                        */
size_t coda_furthest_point_sampling_workspace_bytes(int b, int n, int m);
int coda_furthest_point_sampling_f32(const float *xyz, int b, int n, int m,
                                     int32_t *idx, void *workspace,
                                     size_t workspace_bytes, void *stream);
int coda_furthest_point_sampling_opt_f32(const float *xyz, int b, int n, int m,
                                         int32_t *idx, void *workspace,
                                         size_t workspace_bytes, int distance_mode,
                                         int waves, void *stream);

/* The two-workgroup kernel's only failure mode that the reference's single block per
 * scene (sampling_gpu.cu:72-176) does not have: a workgroup whose partner never
 * answers gives up after the poll limit, and the indices of that launch are WRONG
 * from that round on.  The wave that gives up writes a diagnostic word
 * (bit 31 | scene << 16 | round) to pinned host memory and never waits again, so the
 * launch still ends promptly.  Reporting, since a launch is asynchronous:
 *  - every later coda_furthest_point_sampling*_f32 call of the process returns
 *    CODA_ELOST (and launches nothing) until the word has been read with reset = 1;
 *  - coda_fps_lost_partner_events(reset) returns the word (0 = nothing lost); after
 *    the launch's stream has been synchronised it speaks for that launch.
 * SCOPE: this latch is the one piece of mutable state behind this header, and it is
 * PROCESS-wide (one pinned host word per process, shared by every device, stream and
 * thread).  Two independent callers in one process share it: a loss in one caller's
 * launch makes the other caller's next sampling call return CODA_ELOST, and whoever
 * reads with reset = 1 acknowledges for both -- it is a fault latch ("some sampling
 * result of this process is wrong"), not a per-call status.  A caller that must
 * attribute the loss synchronises its own stream and reads the word (reset = 0)
 * before anyone resets it; bits 16..30 name the scene, bits 0..15 the round.
 * coda_furthest_point_sampling_dbg_f32 is the test hook that produces the event:
 * spin_limit (> 0: polls before giving up, 0: default) and drop_half (0 | 1: that
 * workgroup of every pair exits at once; -1: none).  tests/test_ops_gpu.py.        */
unsigned int coda_fps_lost_partner_events(int reset);
int coda_furthest_point_sampling_dbg_f32(const float *xyz, int b, int n, int m,
                                         int32_t *idx, void *workspace,
                                         size_t workspace_bytes, int spin_limit,
                                         int drop_half, void *stream);

/* ---- gather_points / gather_points_grad --------------------------------------
 * out[b,c,j] = points[b,c,idx[b,j]]            src/sampling_gpu.cu:11-33
 * grad_points[b,c,idx[b,j]] += grad_out[b,c,j] src/sampling_gpu.cu:37-60
 * (grad_points (B,C,N) is zeroed by the call; fp32 atomics, so the summation
 *  order for repeated indices is unspecified, as in the reference).         */
int coda_gather_points_f32(const float *points, const int32_t *idx, float *out,
                           int b, int c, int n, int m, void *stream);
int coda_gather_points_grad_f32(const float *grad_out, const int32_t *idx,
                                float *grad_points, int b, int c, int n, int m,
                                void *stream);

/* Deterministic forms of the two scatter-add adjoints (gather_points_grad here, group_points_grad below): the same
 * sums, but reproducible bit for bit from run to run.  The float atomics of the plain entry points -- and of the
 * reference, sampling_gpu.cu:37-60 / group_points_gpu.cu:46-67 -- add colliding gradients in whatever order the
 * hardware serves them.  Here every addend goes to 64-bit fixed point at a scale taken from the largest magnitude of
 * its own (scene, channel) ROW (62 - ceil(log2(entries + 1)) bits below it: 47 at 32 768 entries per scene, never
 * fewer than 31, i.e. finer than float32 -- per row since round 6, so channels of very different gradient size keep
 * their own relative accuracy), is accumulated with integer atomics (associative: any order, same bits) and
 * converted back once.  `workspace`: device memory, 8-byte aligned, >= coda_scatter_add_det_workspace_bytes(b, c, n)
 * (8 bytes per output element + a 4-byte word per row, rounded up to 256); CODA_ENOSPC otherwise.  A non-finite
 * gradient makes its (scene, channel) row NaN and leaves the other rows untouched.
 * The Python binding uses these by default (CODA_SCATTER=atomic: the plain ones).                                  */
size_t coda_scatter_add_det_workspace_bytes(int b, int c, int n);
int coda_gather_points_grad_det_f32(const float *grad_out, const int32_t *idx,
                                    float *grad_points, int b, int c, int n, int m,
                                    void *workspace, size_t workspace_bytes, void *stream);
int coda_group_points_grad_det_f32(const float *grad_out, const int32_t *idx,
                                   float *grad_points, int b, int c, int n, int npoints,
                                   int nsample, void *workspace, size_t workspace_bytes,
                                   void *stream);

/* ---- ball_query ---------------------------------------------------------------
 * Replaces ball_query(new_xyz (B,M,3), xyz (B,N,3), radius, nsample) -> (B,M,S) i32
 *   wrapper src/ball_query.cpp:11-35, kernel src/ball_query_gpu.cu:12-57.
 * Row j = the nsample smallest point indices k (ascending) with
 * d2(new_xyz[j], xyz[k]) < radius*radius (strict, fp32), padded with the first
 * hit; all zeros when the ball is empty.
 * Two routes, identical results.  "grid" (default when `workspace` is given;
 * device, >= coda_ball_query_workspace_bytes): a per-scene cell table built by
 * one launch and queried by a second (csrc/ball_query_grid.hip).  "scan": brute
 * force over the cloud (small clouds; what NULL/0 selects).  (A third, one-launch
 * LDS-tile route was built and measured 2.4x slower in round 3; removed.)
 * coda_ball_query_opt_f32 / coda_query_and_group_xyz_opt_f32: per-call options --
 * distance_mode as above, route (0 auto | 1 grid | 2 scan; tests, A/B;
 * default: env CODA_BQ=auto|grid|scan).                                  */
size_t coda_ball_query_workspace_bytes(int b, int n, int m, int nsample);
int coda_ball_query_f32(const float *new_xyz, const float *xyz, int32_t *idx,
                        int b, int n, int m, float radius, int nsample,
                        void *workspace, size_t workspace_bytes, void *stream);
int coda_ball_query_opt_f32(const float *new_xyz, const float *xyz, int32_t *idx,
                            int b, int n, int m, float radius, int nsample,
                            void *workspace, size_t workspace_bytes,
                            int distance_mode, int route, void *stream);

/* ---- group_points / group_points_grad ------------------------------------------
 * out[b,c,j,k] = points[b,c,idx[b,j,k]]                 src/group_points_gpu.cu:11-42
 * grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k]      src/group_points_gpu.cu:46-78
 * (grad_points zeroed by the call; fp32 atomics -> order unspecified).      */
int coda_group_points_f32(const float *points, const int32_t *idx, float *out,
                          int b, int c, int n, int npoints, int nsample,
                          void *stream);
int coda_group_points_grad_f32(const float *grad_out, const int32_t *idx,
                               float *grad_points, int b, int c, int n,
                               int npoints, int nsample, void *stream);

/* ---- query_and_group_xyz (fused QueryAndGroup, xyz branch) ---------------------
 * Fuses ball_query + grouping of the coordinates + centring (+ optional 1/radius
 * normalisation) of QueryAndGroup.forward (pointnet2_utils.py:331-349):
 *   idx          (B,M,S) i32   as coda_ball_query_f32
 *   grouped_xyz  (B,3,M,S) f32 = (xyz[idx] - new_xyz[j]) [/ radius if normalize & 1]
 * `normalize` bit 1 (value 2) selects the channels-last layout (B,M,S,3) consumed by
 * the fused shared MLP (coda_sa_mlp.h).  The subtraction and the 1/radius scaling are
 * the same fp32 operations torch performs on the GPU (`-=`, then a multiply by the
 * fp32 reciprocal for `/= radius`).                                          */
int coda_query_and_group_xyz_f32(const float *new_xyz, const float *xyz,
                                 int32_t *idx, float *grouped_xyz, int b, int n,
                                 int m, float radius, int nsample, int normalize,
                                 void *workspace, size_t workspace_bytes,
                                 void *stream);
int coda_query_and_group_xyz_opt_f32(const float *new_xyz, const float *xyz,
                                     int32_t *idx, float *grouped_xyz, int b, int n,
                                     int m, float radius, int nsample, int normalize,
                                     void *workspace, size_t workspace_bytes,
                                     int distance_mode, int route, void *stream);

/* ---- three_nn / three_interpolate / three_interpolate_grad ---------------------
 * three_nn: 3 nearest `known` (B,m,3) of each `unknown` (B,n,3); strict `<`
 *   insertion in ascending k (ties -> lowest k); writes squared distances
 *   dist2 (B,n,3) and idx (B,n,3).            src/interpolate_gpu.cu:12-71
 * three_interpolate: out[b,c,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3, rounded as the
 *   distance arithmetic mode says             src/interpolate_gpu.cu:75-115
 * three_interpolate_grad: scatter-add of grad_out*w into zeroed (B,C,m)
 *                                             src/interpolate_gpu.cu:119-158 */
int coda_three_nn_f32(const float *unknown, const float *known, float *dist2,
                      int32_t *idx, int b, int n, int m, void *stream);
int coda_three_interpolate_f32(const float *points, const int32_t *idx,
                               const float *weight, float *out, int b, int c,
                               int m, int n, void *stream);
int coda_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx,
                                    const float *weight, float *grad_points,
                                    int b, int c, int n, int m, void *stream);
/* the same with the distance / interpolation arithmetic mode as an argument (-1: library default) */
int coda_three_nn_opt_f32(const float *unknown, const float *known, float *dist2,
                          int32_t *idx, int b, int n, int m, int distance_mode, void *stream);
int coda_three_interpolate_opt_f32(const float *points, const int32_t *idx,
                                   const float *weight, float *out, int b, int c,
                                   int m, int n, int distance_mode, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_POINTNET2_H */
