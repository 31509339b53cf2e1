// Error reporting + version of the C ABI (include/stdadk.h).
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/stdadk.h"

#include <stdlib.h>

namespace stdadk {
// STDADK_DRY_RUN=1: validation and planning only, no HIP call (common.h)
bool g_dry_run = [] { const char *e = getenv("STDADK_DRY_RUN"); return e && e[0] == '1'; }();
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- per-launch HIP-event profiler (off by default; not thread-safe, one stream at a time)
// Events without the system-scope fence (both records are on the launch stream of one device): a default event's
// record makes the command processor write back / invalidate the caches around the kernel it brackets, which the
// bracket then counts as kernel time.  The events are pooled: nothing is created on the launch path.
bool g_prof_on = false;
struct ProfRec { std::string name; hipEvent_t e0, e1; };
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static bool prof_event(hipEvent_t *e) {
  if (!g_pool.empty()) { *e = g_pool.back(); g_pool.pop_back(); return true; }
  if (hipEventCreateWithFlags(e, hipEventDisableSystemFence) == hipSuccess) return true;
  (void)hipGetLastError();
  return hipEventCreate(e) == hipSuccess;
}

void prof_before(const char *name, hipStream_t st) {
  ProfRec r;
  r.name = name;
  if (!prof_event(&r.e0) || !prof_event(&r.e1)) return;
  (void)hipEventRecord(r.e0, st);
  g_recs.push_back(r);
}
void prof_after(hipStream_t st) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, st);
}
}  // namespace stdadk

extern "C" int stdadk_profile_enable(int32_t on) {
  for (auto &r : stdadk::g_recs) { stdadk::g_pool.push_back(r.e0); stdadk::g_pool.push_back(r.e1); }
  stdadk::g_recs.clear();
  stdadk::g_prof_on = on != 0;
  return 0;
}

// Synchronises the recorded events and writes "name\tms\n" lines (launch order) into buf.
// Returns the number of bytes needed (call again with a bigger buffer if > cap), or < 0 on error.
extern "C" int64_t stdadk_profile_collect(char *buf, int64_t cap) {
  std::string out;
  for (auto &r : stdadk::g_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) return STDADK_E_ARG;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return STDADK_E_ARG;
    char line[32];
    snprintf(line, sizeof(line), "\t%.6f\n", ms);
    out += r.name;
    out += line;
  }
  if (buf && cap > (int64_t)out.size()) memcpy(buf, out.c_str(), out.size() + 1);
  return (int64_t)out.size() + 1;
}

extern "C" int stdadk_abi_version(void) { return STDADK_ABI_VERSION; }
extern "C" const char *stdadk_last_error(void) { return stdadk::g_err; }
