// Library-level entry points of libadvoc_hip.so (version / error strings).
#include "common.h"

namespace advoc {
static thread_local hipError_t g_last_hip_error = hipSuccess;
void note_hip_error(hipError_t e) { g_last_hip_error = e; }
}  // namespace advoc

extern "C" const char* advoc_last_hip_error(void) {
  return hipGetErrorString(advoc::g_last_hip_error);
}

extern "C" int advoc_abi_version(void) { return ADVOC_ABI_VERSION; }

extern "C" const char* advoc_target_arch(void) { return "gfx950"; }

extern "C" const char* advoc_error_string(int code) {
  switch (code) {
    case ADVOC_OK: return "ok";
    case ADVOC_ERR_BAD_SHAPE: return "bad shape: inconsistent or non-positive dimensions";
    case ADVOC_ERR_UNSUPPORTED: return "unsupported: outside what the gfx950 kernels implement";
    case ADVOC_ERR_HIP: return "HIP runtime error (launch failed)";
    case ADVOC_ERR_NULL: return "null pointer";
    default: return "unknown advoc error code";
  }
}
