// Library identity + error plumbing for the C ABI (include/snerf.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace snerf {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}
}  // namespace snerf

extern "C" int snerf_abi_version(void) { return SNERF_ABI_VERSION; }
extern "C" const char* snerf_last_error(void) { return snerf::g_err; }
extern "C" const char* snerf_target_arch(void) { return "gfx950"; }
