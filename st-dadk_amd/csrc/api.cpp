// Error reporting + version of the C ABI (include/stdadk.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/stdadk.h"

namespace stdadk {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace stdadk

extern "C" int stdadk_abi_version(void) { return STDADK_ABI_VERSION; }
extern "C" const char *stdadk_last_error(void) { return stdadk::g_err; }
