// Version / error plumbing of libevae_hip.so (see include/evae_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/evae_hip.h"

namespace evae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace evae

extern "C" int evae_version(void) { return EVAE_ABI_VERSION; }
extern "C" const char* evae_last_error(void) { return evae::g_err; }
