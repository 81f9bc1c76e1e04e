// ptmi_comm_worker.h -- bounded execution of calls that may never return (the RCCL calls of ptmi_film_comm.h).
// Pure C++17, no HIP / RCCL types: tests/comm_worker_main.cpp exercises it under ThreadSanitizer on the CPU.
//
// BoundedJob: one call (or group of calls) on a worker thread.  RUNNING -> DONE by the worker, RUNNING -> ABANDONED by the
// waiter: ONE compare-exchange decides who owns the outcome.  A worker that finds its job abandoned releases what the late
// call made (the `release_late` callback of bounded_start), because nobody is waiting for it any more.
//
// BoundedWorker: ONE long-lived thread per owner makes all of the owner's calls.  Long-lived on purpose: with a non-blocking
// RCCL communicator a call leaves an asynchronous job behind whose bookkeeping lives in the CALLING thread's thread-local
// storage (RCCL's group.cc), so the calling thread has to outlive the communicator -- a thread per call aborted the process
// in the first gather.  A worker whose call never returns is dropped by its owner (bounded_join resets the owner's pointer:
// the next call gets a new worker); it parks or stays blocked until the process ends.  stop() lets an idle worker exit.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ptw {

using clock = std::chrono::steady_clock;

inline void backoff(unsigned& spins) {
  if (++spins < 200) std::this_thread::yield();
  else std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 50 : 500));
}

struct BoundedJob {
  enum : int { RUNNING = 0, DONE = 1, ABANDONED = 2 };
  std::atomic<int> state{RUNNING};
  int result = 0;                   // the call's status code (ncclResult_t for the RCCL jobs)
  std::string failed_call;          // which call of a group failed
  std::vector<void*> made;          // what the call is making (communicators), written by the callee as soon as it has one
};

struct BoundedWorker {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> jobs;
  bool quit = false;
  static void loop(std::shared_ptr<BoundedWorker> self) {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(self->m);
        self->cv.wait(lk, [&] { return self->quit || !self->jobs.empty(); });
        if (self->jobs.empty()) return;
        f = std::move(self->jobs.front());
        self->jobs.pop_front();
      }
      f();
    }
  }
  void post(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(m); jobs.push_back(std::move(f)); }
    cv.notify_one();
  }
  void stop() {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    cv.notify_one();
  }
};

// The owner's worker, created on first use.
inline std::shared_ptr<BoundedWorker> worker_of(std::shared_ptr<BoundedWorker>& slot) {
  if (!slot) {
    slot = std::make_shared<BoundedWorker>();
    std::thread(BoundedWorker::loop, slot).detach();
  }
  return slot;
}

// Run fn(job) -> int on the owner's worker.  release_late(job) runs ON THE WORKER if the call comes back after it was abandoned.
template <class F, class R>
std::shared_ptr<BoundedJob> bounded_start(std::shared_ptr<BoundedWorker>& slot, size_t n_made, F fn, R release_late) {
  auto job = std::make_shared<BoundedJob>();
  job->made.assign(n_made, nullptr);
  worker_of(slot)->post([job, fn, release_late]() mutable {
    job->result = fn(*job);
    int expected = BoundedJob::RUNNING;
    if (!job->state.compare_exchange_strong(expected, BoundedJob::DONE)) release_late(*job);
  });
  return job;
}

// Wait for a job against a deadline and an optional abort request.  true: the job is DONE and its result is the caller's;
// false: it was abandoned -- the worker owns whatever the call still makes, and the owner lets go of that worker.
inline bool bounded_join(std::shared_ptr<BoundedWorker>& slot, BoundedJob& job, clock::time_point deadline,
                         const std::atomic<bool>* abort_request = nullptr) {
  unsigned spins = 0;
  for (;;) {
    if (job.state.load(std::memory_order_acquire) == BoundedJob::DONE) return true;
    if ((abort_request && abort_request->load()) || clock::now() > deadline) {
      int expected = BoundedJob::RUNNING;
      if (!job.state.compare_exchange_strong(expected, BoundedJob::ABANDONED)) return true;   // lost the race: it finished just now
      slot.reset();
      return false;
    }
    backoff(spins);
  }
}

}  // namespace ptw
