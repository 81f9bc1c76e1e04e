// One background thread with join: reference src/AsyncTask.hpp:14-67 (same interface and error behaviour).
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <stdexcept>
#include <thread>

#include "logging.hpp"

class AsyncTask {
public:
  AsyncTask() : running(false) {}
  virtual ~AsyncTask() { waitForCompletion(); }

  /// Run a function in a new thread.  Throws std::logic_error if a task is already in progress.
  void run(std::function<void()>&& f) {
    if (job != nullptr) {
      auto error = "Attempted to run AsyncTask while a job was in progress.";
      pt_log::error_(error);
      throw std::logic_error(error);
    }
    asyncFunc = std::move(f);
    job.reset(new std::thread([this]() {
      running = true;
      asyncFunc();
      running = false;
    }));
  }

  /// Wait for the job to complete.
  void waitForCompletion() {
    if (job != nullptr) {
      try {
        job->join();
        job.reset();
      } catch (std::system_error&) {
        pt_log::error_("Thread could not be joined.");
      }
    }
  }

  bool isRunning() const { return running; }

private:
  std::function<void()> asyncFunc;
  std::unique_ptr<std::thread> job;
  std::atomic<bool> running;
};
