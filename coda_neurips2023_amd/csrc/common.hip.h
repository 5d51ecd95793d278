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

// Squared distance in the reference's source order, one rounding per
// operation.  The translation units are compiled with -ffp-contract=off; the
// explicit __f*_rn intrinsics make the contract independent of flags.
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ int wave_id() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
}
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace coda
