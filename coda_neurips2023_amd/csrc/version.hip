// version.hip -- library identification for libcoda_hip.so.
#include "common.hip.h"

CODA_API const char *coda_version(void) { return "coda_hip gfx950 abi1"; }
