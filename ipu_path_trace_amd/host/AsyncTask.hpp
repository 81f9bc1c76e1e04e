// Runs one host job at a time on a background thread while the device works on the next step.
// Interface and error behaviour of the reference's AsyncTask (src/AsyncTask.hpp:14-67): run() throws
// std::logic_error when a job is still pending, waitForCompletion() joins it, isRunning() reports progress.
#pragma once
#include <atomic>
#include <exception>
#include <functional>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <utility>

#include "logging.hpp"

class AsyncTask {
public:
  AsyncTask() = default;
  AsyncTask(const AsyncTask&) = delete;
  AsyncTask& operator=(const AsyncTask&) = delete;
  virtual ~AsyncTask() {
    try { waitForCompletion(); } catch (...) {}   // a destructor must not throw; the job's failure was not collected
  }

  void run(std::function<void()>&& f) {
    if (worker.joinable()) {
      static const char* msg = "Attempted to run AsyncTask while a job was in progress.";
      pt_log::error_(msg);
      throw std::logic_error(msg);
    }
    if (failure) {   // (cannot happen through waitForCompletion, which collects it; never let a stale failure reach a later job)
      pt_log::error_("AsyncTask: an earlier job's failure was never collected; dropping it.");
      failure = nullptr;
    }
    task = std::move(f);
    busy.store(true);
    worker = std::thread([this] {
      // an exception must not escape the thread (std::terminate): keep it and rethrow it to whoever joins
      try { task(); } catch (...) { failure = std::current_exception(); }
      busy.store(false);
    });
  }

  void waitForCompletion() {
    if (!worker.joinable()) return;
    try {
      worker.join();
    } catch (const std::system_error&) {
      pt_log::error_("Thread could not be joined.");
    }
    if (failure) {
      std::exception_ptr e = failure;
      failure = nullptr;
      std::rethrow_exception(e);
    }
  }

  bool isRunning() const { return busy.load(); }

private:
  std::function<void()> task;
  std::thread worker;
  std::atomic<bool> busy{false};
  std::exception_ptr failure;
};
