// Exercises ipu_path_trace_amd/csrc/ptmi_comm_worker.h -- the machinery that bounds RCCL calls which may never return -- with
// stand-in calls, under ThreadSanitizer (tests/test_comm_worker.py builds it with -fsanitize=thread).  Scenarios:
//  1. a call that returns in time: its result and what it made belong to the caller; the same worker THREAD serves the next call
//     (thread-local state of the callee survives between calls: what a non-blocking RCCL communicator needs);
//  2. a call that blocks past the deadline: bounded_join gives up on time, the owner's worker slot is emptied, the next call
//     gets a NEW thread and is not held up by the stuck one;
//  3. ... and when the stuck call finally returns, the worker releases what it made (release_late), exactly once;
//  4. an abort request from another thread ends the wait at once;
//  5. a call finishing at the very moment of the deadline is owned by exactly one side (hammered);
//  6. stop() lets an idle worker exit.
#include <cassert>
#include <cstdio>
#include <set>

#include "ptmi_comm_worker.h"

using namespace ptw;
using ms = std::chrono::milliseconds;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); ++failures; } } while (0)

int main() {
  auto no_release = [](BoundedJob&) {};
  {  // 1
    std::shared_ptr<BoundedWorker> slot;
    thread_local int tls_calls = 0;
    std::thread::id first;
    auto j1 = bounded_start(slot, 1, [&](BoundedJob& j) { first = std::this_thread::get_id(); tls_calls += 1; j.made[0] = &failures; return 7; }, no_release);
    CHECK(bounded_join(slot, *j1, clock::now() + ms(2000)));
    CHECK(j1->result == 7 && j1->made[0] == &failures && slot);
    int seen = 0;
    std::thread::id second;
    auto j2 = bounded_start(slot, 0, [&](BoundedJob&) { second = std::this_thread::get_id(); tls_calls += 1; seen = tls_calls; return 0; }, no_release);
    CHECK(bounded_join(slot, *j2, clock::now() + ms(2000)));
    CHECK(first == second && seen == 2);                       // one long-lived thread, its thread-local state intact
    slot->stop();
  }
  std::atomic<int> released{0};
  {  // 2 + 3
    std::shared_ptr<BoundedWorker> slot;
    std::atomic<bool> let_go{false};
    std::atomic<size_t> stuck_thread{0}, fresh_thread{0};   // (hashes of the ids: read while the stuck call is still running)
    auto stuck = bounded_start(slot, 1, [&](BoundedJob& j) {
      stuck_thread.store(std::hash<std::thread::id>()(std::this_thread::get_id()));
      j.made[0] = &released;                                   // "a communicator" that exists by the time the call returns
      while (!let_go.load()) std::this_thread::sleep_for(ms(1));
      return 0;
    }, [&](BoundedJob& j) { if (j.made[0]) released.fetch_add(1); });
    const auto t0 = clock::now();
    CHECK(!bounded_join(slot, *stuck, t0 + ms(150)));          // abandoned ...
    const auto waited = std::chrono::duration_cast<ms>(clock::now() - t0).count();
    CHECK(waited >= 150 && waited < 1000);                     // ... on time
    CHECK(!slot);                                              // the owner has let go of the stuck worker
    auto next = bounded_start(slot, 0, [&](BoundedJob&) { fresh_thread.store(std::hash<std::thread::id>()(std::this_thread::get_id())); return 3; }, no_release);
    CHECK(bounded_join(slot, *next, clock::now() + ms(2000)) && next->result == 3);
    CHECK(fresh_thread.load() != stuck_thread.load() && stuck_thread.load() != 0);                       // a new worker, not queued behind the stuck call
    CHECK(released.load() == 0);
    let_go.store(true);                                        // the "peer" finally arrives
    for (int i = 0; i < 2000 && released.load() == 0; ++i) std::this_thread::sleep_for(ms(1));
    CHECK(released.load() == 1);                               // the late arrival was released by the worker, once
    slot->stop();
  }
  {  // 4
    std::shared_ptr<BoundedWorker> slot;
    auto let_go = std::make_shared<std::atomic<bool>>(false);   // (outlives this scope: the abandoned call is still polling it)
    std::atomic<bool> abort_request{false};
    auto stuck = bounded_start(slot, 0, [let_go](BoundedJob&) { while (!let_go->load()) std::this_thread::sleep_for(ms(1)); return 0; }, no_release);
    std::thread other([&] { std::this_thread::sleep_for(ms(50)); abort_request.store(true); });
    const auto t0 = clock::now();
    CHECK(!bounded_join(slot, *stuck, t0 + ms(60000), &abort_request));
    CHECK(std::chrono::duration_cast<ms>(clock::now() - t0).count() < 5000);
    other.join();
    let_go->store(true);
  }
  {  // 5: completion racing the deadline -- exactly one owner, whatever the interleaving
    int both = 0, neither = 0;
    for (int round = 0; round < 300; ++round) {
      std::shared_ptr<BoundedWorker> slot;
      auto late = std::make_shared<std::atomic<int>>(0);        // (the abandoned call may still be running when this round ends)
      const int nap = 200 + (round % 7) * 50;
      auto j = bounded_start(slot, 0, [nap](BoundedJob&) { std::this_thread::sleep_for(std::chrono::microseconds(nap)); return 0; },
                             [late](BoundedJob&) { late->fetch_add(1); });
      const bool mine = bounded_join(slot, *j, clock::now() + std::chrono::microseconds(350));
      std::this_thread::sleep_for(ms(2));
      const int state = j->state.load();
      if (mine && late->load()) both += 1;
      if (!mine && state != BoundedJob::ABANDONED) neither += 1;
      CHECK(mine ? state == BoundedJob::DONE : state == BoundedJob::ABANDONED);
      if (slot) slot->stop();
    }
    CHECK(both == 0 && neither == 0);
  }
  {  // 6
    std::shared_ptr<BoundedWorker> slot;
    auto j = bounded_start(slot, 0, [](BoundedJob&) { return 0; }, no_release);
    CHECK(bounded_join(slot, *j, clock::now() + ms(2000)));
    std::weak_ptr<BoundedWorker> w = slot;
    slot->stop();
    slot.reset();
    for (int i = 0; i < 2000 && !w.expired(); ++i) std::this_thread::sleep_for(ms(1));
    CHECK(w.expired());                                        // the thread has returned and dropped the last reference
  }
  std::this_thread::sleep_for(ms(50));
  std::printf(failures ? "COMM_WORKER_FAILED %d\n" : "COMM_WORKER_OK\n", failures);
  return failures ? 1 : 0;
}
