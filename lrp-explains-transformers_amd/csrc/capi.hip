// capi.hip -- library identity + per-thread error slot of the C ABI (include/lrp_hip.h).
#include "common.hpp"

thread_local int g_lrp_last_hip_error = 0;

extern "C" int lrp_version(void) { return 8; }
extern "C" const char* lrp_build_arch(void) { return "gfx950"; }
extern "C" int lrp_last_hip_error(void) { return g_lrp_last_hip_error; }
