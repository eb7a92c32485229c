// Thread-local error string + ABI version for libradmmm_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/radmmm_hip.h"

namespace radmmm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Experiment / test switches (tile overrides, forced kernel choices, A/B paths) are honoured only under RADMMM_DEBUG=1,
// which is read ONCE per process: a production launch never calls getenv.  Supported switches (RADMMM_GEMM_CUS,
// RADMMM_PRECISION, RADMMM_CHECK_SATURATION, RADMMM_LIB_PATH) do not go through here; see INTEGRATION.md.
const char* debug_env(const char* name) {
  static const bool on = [] {
    const char* e = getenv("RADMMM_DEBUG");
    return e && atoi(e) != 0;
  }();
  return on ? getenv(name) : nullptr;
}
}  // namespace radmmm

extern "C" const char* radmmm_last_error(void) { return radmmm::g_err; }
extern "C" int radmmm_abi_version(void) { return RADMMM_ABI_VERSION; }
