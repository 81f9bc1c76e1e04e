// Minimal stand-in for the reference's spdlog logger (src/ipu_utils.hpp:25-28, src/main.cpp:71-90):
// same level names, same "[HH:MM:SS.us] [L] [tid] message" pattern, {}-style placeholders.
#pragma once
#include <chrono>
#include <cstdio>
#include <ctime>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>

namespace pt_log {

enum Level { trace = 0, debug, info, warn, err, critical, off };

inline Level& level() { static Level l = info; return l; }

inline void setLevel(const std::string& name) {
  static const char* names[] = {"trace", "debug", "info", "warn", "err", "critical", "off"};
  for (int i = 0; i < 7; ++i) if (name == names[i]) { level() = (Level)i; return; }
  throw std::runtime_error("Invalid log-level: '" + name + "'");  // main.cpp:83-87
}

inline void format_into(std::ostringstream& os, const char* fmt) { os << fmt; }
template <class T, class... Rest>
void format_into(std::ostringstream& os, const char* fmt, const T& v, const Rest&... rest) {
  for (; *fmt; ++fmt) {
    if (fmt[0] == '{' && fmt[1] == '}') { os << v; format_into(os, fmt + 2, rest...); return; }
    os << *fmt;
  }
}

template <class... Args>
void log(Level l, const char* fmt, const Args&... args) {
  if (l < level()) return;
  static std::mutex m;
  std::ostringstream os;
  format_into(os, fmt, args...);
  auto now = std::chrono::system_clock::now();
  std::time_t t = std::chrono::system_clock::to_time_t(now);
  auto us = std::chrono::duration_cast<std::chrono::microseconds>(now.time_since_epoch()).count() % 1000000;
  std::tm tm;
  localtime_r(&t, &tm);
  static const char tags[] = "TDIWEC";
  std::lock_guard<std::mutex> lock(m);
  std::fprintf(stdout, "[%02d:%02d:%02d.%06ld] [%c] [%zu] %s\n", tm.tm_hour, tm.tm_min, tm.tm_sec, (long)us, tags[l],
               std::hash<std::thread::id>()(std::this_thread::get_id()) % 100000, os.str().c_str());
  std::fflush(stdout);
}

template <class... A> void info_(const char* f, const A&... a) { log(info, f, a...); }
template <class... A> void debug_(const char* f, const A&... a) { log(debug, f, a...); }
template <class... A> void warn_(const char* f, const A&... a) { log(warn, f, a...); }
template <class... A> void error_(const char* f, const A&... a) { log(err, f, a...); }
template <class... A> void trace_(const char* f, const A&... a) { log(trace, f, a...); }

}  // namespace pt_log
