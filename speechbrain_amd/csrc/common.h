// Shared host-side helpers for the C-ABI entry points (argument checks, error slot).
#pragma once
#include <sbk_device.h>

#include "sbk.h"

namespace sbk {

// Records a message in the thread-local error slot and returns `code`.
int fail(int code, const char* fmt, ...);
// Returns 0 or the hipError_t of the launch that just happened (and records it).
int launch_status(const char* what);

// Optional per-launch timing with HIP events on the launch stream (sbk_prof_* in include/sbk.h).
// Costs nothing when disabled.  `flops` / `bytes` are the ALGORITHMIC work of the launch.
struct ProfScope {
  ProfScope(const char* name, double flops, double bytes, hipStream_t st);
  ~ProfScope();
  int slot;
  hipStream_t st;
};

bool prof_enabled();  // per-launch timing is on (sbk_prof_enable)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline hipStream_t as_stream(sbk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace sbk

#define SBK_EINVAL (-22)
#define SBK_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return sbk::fail(SBK_EINVAL, __VA_ARGS__);     \
  } while (0)
