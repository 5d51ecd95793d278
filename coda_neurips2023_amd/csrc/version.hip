// version.hip -- library identification, the per-call options record and the default distance-arithmetic mode.
#include "common.hip.h"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace coda {
namespace {
int initial_mode() {
  const char *e = getenv("CODA_DISTANCE_MODE");
  if (e && e[0] >= '0' && e[0] < '0' + kDistanceModes && e[1] == 0) return e[0] - '0';
  return kDefaultDistanceMode;
}
thread_local CallOptions tls_call_options;
}  // namespace

CallOptions &call_options() { return tls_call_options; }

int default_distance_mode() {
  static const int mode = initial_mode();  // read once
  return mode;
}

int distance_mode() {
  const int m = tls_call_options.distance_mode;
  return m >= 0 && m < kDistanceModes ? m : default_distance_mode();
}
}  // namespace coda

namespace coda {
int raise_dynamic_lds(const void *kernel, size_t bytes, size_t static_bytes) {
  if (bytes + static_bytes <= 64 * 1024) return CODA_OK;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = -1;
  }
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> raised;
  const auto key = std::make_pair(kernel, dev);
  if (dev >= 0) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = raised.find(key);
    if (it != raised.end() && it->second >= bytes) return CODA_OK;
  }
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return CODA_ENOSPC;
  }
  if (dev >= 0) {
    std::lock_guard<std::mutex> lock(mu);
    size_t &slot = raised[key];
    if (slot < bytes) slot = bytes;
  }
  return CODA_OK;
}
}  // namespace coda

CODA_API const char *coda_version(void) { return "coda_hip gfx950 abi3"; }

CODA_API int coda_get_distance_mode(void) { return coda::default_distance_mode(); }
