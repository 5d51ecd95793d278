// dropout.hip.h -- counter-based dropout decision shared by the token-wise kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace coda {

// keep-probability threshold on 24 bits: keep iff hash24 >= thresh24 (0 = dropout off)
inline uint32_t drop_thresh24(float p) {
  if (!(p > 0.f)) return 0u;
  const double t = static_cast<double>(p) * 16777216.0;
  return t >= 16777215.0 ? 16777215u : static_cast<uint32_t>(t + 0.5);
}

__device__ __forceinline__ uint32_t fold_seed(uint32_t seed, const uint64_t *seed_dev) {
  if (!seed_dev) return seed;
  const uint64_t v = *seed_dev;
  return seed ^ static_cast<uint32_t>(v) ^ static_cast<uint32_t>(v >> 32) * 0x9E3779B9u;
}

// 32-bit finaliser (two multiply-xorshift rounds) of (element index ^ seed)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool keep_elem(uint32_t seed, uint32_t idx, uint32_t thresh24) {
  return (mix32(idx ^ seed) >> 8) >= thresh24;
}

}  // namespace coda
