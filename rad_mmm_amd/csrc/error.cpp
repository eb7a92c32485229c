// Thread-local error string + ABI version for libradmmm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/radmmm_hip.h"

namespace radmmm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace radmmm

extern "C" const char* radmmm_last_error(void) { return radmmm::g_err; }
extern "C" int radmmm_abi_version(void) { return RADMMM_ABI_VERSION; }
