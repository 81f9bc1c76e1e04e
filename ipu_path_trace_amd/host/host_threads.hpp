// Team size of the host-side OpenMP loops (film accumulate, worklist bookkeeping).
//
// They are memory-bound passes over ~1 M records: a few threads saturate them, while a team as wide as a
// 128-core host leaves spinning workers that slow the main thread's worklist transfers (measured on the MI355X
// box: 5-25 ms per 275 ms step against 1-2 ms with a small team).  OMP_NUM_THREADS still decides if it is set.
// The clause is needed on every loop: omp_set_num_threads() only affects the calling thread, and these loops run
// on the AsyncTask thread.
#pragma once
#include <omp.h>

#include <algorithm>
#include <cstdlib>

inline int hostLoopThreads() {
  static const int n = [] {
    if (const char* e = std::getenv("OMP_NUM_THREADS")) { const int v = std::atoi(e); if (v > 0) return v; }
    return std::min(8, omp_get_num_procs());
  }();
  return n;
}
