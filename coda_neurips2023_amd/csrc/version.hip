// version.hip -- library identification and the process-wide distance-arithmetic mode.
#include "common.hip.h"

#include <atomic>
#include <cstdlib>

namespace coda {
namespace {
int initial_mode() {
  const char *e = getenv("CODA_DISTANCE_MODE");
  if (e && e[0] >= '0' && e[0] < '0' + kDistanceModes && e[1] == 0) return e[0] - '0';
  return kDefaultDistanceMode;
}
std::atomic<int> g_distance_mode{-1};
}  // namespace

int distance_mode() {
  int m = g_distance_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    m = initial_mode();
    g_distance_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
}  // namespace coda

CODA_API const char *coda_version(void) { return "coda_hip gfx950 abi2"; }

CODA_API int coda_set_distance_mode(int mode) {
  if (mode < 0 || mode >= coda::kDistanceModes) return CODA_EINVAL;
  coda::g_distance_mode.store(mode, std::memory_order_relaxed);
  return CODA_OK;
}

CODA_API int coda_get_distance_mode(void) { return coda::distance_mode(); }
