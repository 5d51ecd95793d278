// common.hip.h -- shared helpers for the gfx950 kernels of libcoda_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "coda_pointnet2.h"

#define CODA_API extern "C" __attribute__((visibility("default")))

namespace coda {

constexpr int kWave = 64;  // CDNA wavefront width (hard-coded: gfx950 only)

// Launch-status helper: the C ABI returns hipError_t values instead of the
// reference's print + exit(-1) (include/cuda_utils.h:32-41).
// hipGetLastError() is sticky per thread: clear what earlier, unrelated runtime
// calls (e.g. the framework's device probing) may have left behind before a
// launch whose status is going to be reported.
inline void clear_sticky_error() { (void)hipGetLastError(); }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CODA_OK : static_cast<int>(e);
}

// ---- distance arithmetic mode (include/coda_pointnet2.h, "Arithmetic contract") ----
// The reference evaluates `dx*dx + dy*dy + dz*dz` (sampling_gpu.cu:103-107,
// ball_query_gpu.cu:34-35, interpolate_gpu.cu:36) and `p1*w1 + p2*w2 + p3*w3`
// (interpolate_gpu.cu:101-102) in a binary built by nvcc with the default
// -fmad=true (setup.py:26-28), i.e. with FMA contraction.  Which contraction is
// not recoverable without nvcc, so the kernels implement all three candidates
// and the mode is selected per call (the *_opt entry points; default CODA_DISTANCE_MODE):
//   0  no contraction:            (a*a' + b*b') + c*c'      one rounding per operation
//   1  fma(c,c', fma(a,a', b*b'))  (first product of the inner sum contracted: LLVM/NVVM order)
//   2  fma(c,c', fma(b,b', a*a'))
// The translation units are compiled with -ffp-contract=off; the explicit
// __f*_rn intrinsics make every mode independent of compiler flags.
constexpr int kDistanceModes = 3;
constexpr int kDefaultDistanceMode = 1;
// ---- per-call options -------------------------------------------------------------------------------------------
// The library keeps NO mutable process-wide state (the reference's ops are stateless): every switch has a
// library default -- an environment variable read once -- and the *_opt entry points take the value as an argument.
// An *_opt entry point installs its arguments in this thread-local record for the duration of the call; the plain
// entry points and the kernels' launch code read it (an unset field = the library default), so concurrent calls from
// different threads with different options never see each other, and an entry point that issues other entry points
// (csrc/decoder_stack.hip) passes its options on without extra parameters.
struct CallOptions {
  int distance_mode = -1;  // -1 default | 0 | 1 | 2
  int fps_waves = 0;       // 0 default | 8 | 16
  int fps_spin_limit = 0;  // coda_furthest_point_sampling_dbg_f32 only: polls before a partner counts as lost (0: default)
  int fps_drop_half = -1;  // coda_furthest_point_sampling_dbg_f32 only: the workgroup of the pair that exits at once
  int bq_route = 0;        // 0 auto | 1 grid | 2 scan
  int mfma_dtype = -1;     // -1 default | 0 fp32 | 1 bf16 | 2 bf16x3
  void *attn_ds_ws = nullptr;  // coda_mha_bwd_ws_f32: the caller's dS workspace (coda_attention.h)
  size_t attn_ds_bytes = 0;
};
CallOptions &call_options();  // version.hip: thread_local
struct ScopedCallOptions {
  CallOptions saved;
  explicit ScopedCallOptions(const CallOptions &o) : saved(call_options()) { call_options() = o; }
  ~ScopedCallOptions() { call_options() = saved; }
  ScopedCallOptions(const ScopedCallOptions &) = delete;
  ScopedCallOptions &operator=(const ScopedCallOptions &) = delete;
};
int distance_mode();          // this call's mode: the option if set, else the library default
int default_distance_mode();  // CODA_DISTANCE_MODE (0|1|2) if set, else 1

template <int DM>
__device__ __forceinline__ float dot3(float a, float a2, float b, float b2, float c, float c2) {
  if constexpr (DM == 1) return __fmaf_rn(c, c2, __fmaf_rn(a, a2, __fmul_rn(b, b2)));
  else if constexpr (DM == 2) return __fmaf_rn(c, c2, __fmaf_rn(b, b2, __fmul_rn(a, a2)));
  else return __fadd_rn(__fadd_rn(__fmul_rn(a, a2), __fmul_rn(b, b2)), __fmul_rn(c, c2));
}
template <int DM>
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
  return dot3<DM>(dx, dx, dy, dy, dz, dz);
}

// CODA_DISPATCH_DM(mode, stmt): runs `stmt` with `DM` bound to the compile-time mode.
#define CODA_DISPATCH_DM(mode, ...)                              \
  switch (mode) {                                                \
    case 0: { constexpr int DM = 0; __VA_ARGS__; } break;        \
    case 2: { constexpr int DM = 2; __VA_ARGS__; } break;        \
    default: { constexpr int DM = 1; __VA_ARGS__; } break;       \
  }

__device__ __forceinline__ int wave_id() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
}
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Opt a kernel into more than 64 KB of dynamic LDS on the CURRENT device (version.hip).  Remembered per
// (kernel entry point, device), thread-safe, so a process that drives several GPUs raises the limit on each of
// them and the steady-state launch path makes no runtime call.  A refused opt-in is reported as CODA_ENOSPC
// ("does not fit"), which callers with a fallback route (matcher solver="auto") act on.
// `static_bytes`: the kernel's static LDS, which counts against the same 64 KB default.
int raise_dynamic_lds(const void *kernel, size_t bytes, size_t static_bytes);
template <typename K>
inline int raise_dynamic_lds(K kernel, size_t bytes, size_t static_bytes = 0) {
  return raise_dynamic_lds(reinterpret_cast<const void *>(kernel), bytes, static_bytes);
}
// launch errors that mean "this launch configuration does not fit the device"
inline int launch_status_nospace() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return CODA_OK;
  if (e == hipErrorInvalidValue || e == hipErrorInvalidConfiguration || e == hipErrorLaunchOutOfResources ||
      e == hipErrorOutOfMemory)
    return CODA_ENOSPC;
  return static_cast<int>(e);
}

}  // namespace coda
