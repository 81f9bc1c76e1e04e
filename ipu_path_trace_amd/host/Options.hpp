// Parsed command line: the role boost::program_options::variables_map plays in the reference
// (src/main.cpp:39-60, PathTracerApp.cpp:794-830).  Boost is not available on this platform.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>

/// option name -> value text
struct OptionMap {
  std::map<std::string, std::string> values;
  bool has(const std::string& k) const { return values.count(k) != 0; }
  const std::string& str(const std::string& k) const {
    auto it = values.find(k);
    if (it == values.end()) throw std::runtime_error("the option '--" + k + "' is required but missing");
    return it->second;
  }
  std::uint32_t u32(const std::string& k) const { return (std::uint32_t)std::stoul(str(k)); }
  std::uint64_t u64(const std::string& k) const { return std::stoull(str(k)); }
  float f32(const std::string& k) const { return std::stof(str(k)); }
  bool flag(const std::string& k) const { return has(k) && str(k) == "true"; }
};

struct OptionSpec {
  std::string name;      // long name
  char shortName;        // 0 if none
  std::string defaultValue;
  bool required, isSwitch;
  std::string help;
};
