// Named host-phase ranges for a profiler's timeline: the counterpart of the reference's pvti::Tracepoint calls around every
// host phase (src/PathTracerApp.cpp:313-323, :567-640, :681-705, :718-763; the range names are the reference's own where the
// phase exists there).  ROCTx ranges, resolved at run time with dlopen -- librocprofiler-sdk-roctx (what rocprofv3
// --marker-trace records), else the older libroctx64 -- so the host has no link-time dependency on either and runs unchanged
// where neither exists (the ranges are then no-ops).
#pragma once
#include <dlfcn.h>

namespace pt_trace {

struct Api {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  const char* library = "";
  Api() {
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
      if (push && pop) { library = name; return; }
      push = nullptr; pop = nullptr;
    }
  }
};

inline const Api& api() {
  static const Api a;   // resolved once, thread-safe
  return a;
}

/// Scoped range on the calling thread (ROCTx keeps one stack of ranges per thread), as pvti::Tracepoint's scoped form.
class Range {
public:
  explicit Range(const char* name) : active(api().push != nullptr) { if (active) api().push(name); }
  ~Range() { if (active) api().pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
private:
  bool active;
};

}  // namespace pt_trace
