#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace sbk {
namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return static_cast<int>(e);
}
}  // namespace sbk

// ---------------------------------------------------------------- event profiler
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

namespace sbk {
namespace {
struct ProfRec {
  const char* name;
  double flops, bytes;
  hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

bool prof_enabled() { return g_prof_on; }

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t s) : slot(-1), st(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{name, flops, bytes, take_event(), take_event()};
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, st);
  slot = (int)g_prof.size();
  g_prof.push_back(r);
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[slot].e1, st);
}
}  // namespace sbk

extern "C" void sbk_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(sbk::g_prof_mu);
  sbk::g_prof_on = on != 0;
}

extern "C" void sbk_prof_reset(void) {
  std::lock_guard<std::mutex> lk(sbk::g_prof_mu);
  for (auto& r : sbk::g_prof) {
    sbk::g_pool.push_back(r.e0);
    sbk::g_pool.push_back(r.e1);
  }
  sbk::g_prof.clear();
}

// Writes one line per kernel class: "name count total_ms flops bytes\n".  Synchronises on the
// recorded events.  Returns the number of bytes needed (call again with a larger buffer if > cap).
extern "C" size_t sbk_prof_report(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(sbk::g_prof_mu);
  struct Agg {
    const char* name;
    long count;
    double ms, flops, bytes;
  };
  std::vector<Agg> agg;
  for (auto& r : sbk::g_prof) {
    float ms = 0.0f;
    (void)hipEventSynchronize(r.e1);
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    Agg* a = nullptr;
    for (auto& x : agg)
      if (strcmp(x.name, r.name) == 0) a = &x;
    if (!a) {
      agg.push_back(Agg{r.name, 0, 0, 0, 0});
      a = &agg.back();
    }
    a->count++;
    a->ms += ms;
    a->flops += r.flops;
    a->bytes += r.bytes;
  }
  std::string out;
  char line[256];
  for (auto& x : agg) {
    snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", x.name, x.count, x.ms, x.flops, x.bytes);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

extern "C" int sbk_abi_version(void) { return SBK_ABI_VERSION; }
extern "C" const char* sbk_last_error(void) { return sbk::g_err; }
