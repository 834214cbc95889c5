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

extern "C" int sbk_abi_version(void) { return SBK_ABI_VERSION; }
extern "C" const char* sbk_last_error(void) { return sbk::g_err; }
