// Which RCCL call blocks when a peer is missing?  A lone rank 0 of a 2-rank NON-BLOCKING communicator.
// build: hipcc -O1 -o scripts/diag/rccl_nonblocking scripts/diag/rccl_nonblocking.cpp -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

static double now() {
  using namespace std::chrono;
  static const auto t0 = steady_clock::now();
  return duration<double>(steady_clock::now() - t0).count();
}

int main(int argc, char** argv) {
  const int nranks = argc > 1 ? atoi(argv[1]) : 2;
  int ver = 0;
  ncclGetVersion(&ver);
  printf("[%.2f] rccl runtime version %d, compiled %d\n", now(), ver, NCCL_VERSION_CODE);
  hipSetDevice(0);
  ncclUniqueId id;
  printf("[%.2f] ncclGetUniqueId -> %d\n", now(), (int)ncclGetUniqueId(&id));
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  cfg.blocking = 0;
  ncclComm_t comm = nullptr;
  std::atomic<int> init_done{0};
  std::atomic<int> init_rc{-1};
  printf("[%.2f] calling ncclCommInitRankConfig(nranks=%d, rank 0, blocking=0) on a helper thread\n", now(), nranks);
  fflush(stdout);
  std::thread t([&]() {
    hipSetDevice(0);
    const ncclResult_t r = ncclCommInitRankConfig(&comm, nranks, id, 0, &cfg);
    init_rc = (int)r;
    init_done = 1;
    printf("[%.2f] ncclCommInitRankConfig returned %d (%s), comm=%p\n", now(), (int)r, ncclGetErrorString(r), (void*)comm);
    fflush(stdout);
  });
  for (int i = 0; i < 60 && !init_done; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
  printf("[%.2f] after 6 s: init call %s\n", now(), init_done ? "has returned" : "IS STILL BLOCKED");
  if (init_done && comm) {
    for (int i = 0; i < 30; ++i) {
      ncclResult_t st = ncclSuccess;
      const ncclResult_t q = ncclCommGetAsyncError(comm, &st);
      if (i % 10 == 0) printf("[%.2f] ncclCommGetAsyncError -> %d, state %d (%s)\n", now(), (int)q, (int)st, ncclGetErrorString(st));
      if (st != ncclInProgress) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    std::atomic<int> abort_done{0};
    printf("[%.2f] calling ncclCommAbort on a helper thread\n", now());
    fflush(stdout);
    std::thread a([&]() {
      const ncclResult_t r = ncclCommAbort(comm);
      abort_done = 1;
      printf("[%.2f] ncclCommAbort returned %d (%s)\n", now(), (int)r, ncclGetErrorString(r));
      fflush(stdout);
    });
    for (int i = 0; i < 100 && !abort_done; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    printf("[%.2f] after 10 s: abort %s\n", now(), abort_done ? "has returned" : "IS STILL BLOCKED");
    fflush(stdout);
    if (abort_done) a.join(); else a.detach();
  }
  fflush(stdout);
  if (init_done) t.join(); else t.detach();
  printf("[%.2f] exiting\n", now());
  fflush(stdout);
  _Exit(0);
}
