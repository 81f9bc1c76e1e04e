// Application object: reference src/PathTracerApp.hpp:36-57.  init() / execute() / addToolOptions() keep
// their roles; build() (Poplar graph construction) has no counterpart -- the device program is libptmi.so.
#pragma once
#include <chrono>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "AccumulatedImage.hpp"
#include "IpuPathTraceJob.hpp"
#include "LoadBalancer.hpp"
#include "NifModel.hpp"
#include "ptmi.h"

/// Parsed command line: option name -> value text (the role boost::program_options::variables_map plays).
struct OptionMap {
  std::map<std::string, std::string> values;
  bool has(const std::string& k) const { return values.count(k) != 0; }
  const std::string& str(const std::string& k) const;
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

struct PathTracerState {
  PathTracerState(std::uint32_t imageWidth, std::uint32_t imageHeight, std::size_t workItems)
      : work(workItems), film(imageWidth, imageHeight) {}
  LoadBalancer work;
  AccumulatedImage film;
};

struct PathTracerApp {
  PathTracerApp();
  virtual ~PathTracerApp();

  /// All initialisation that needs no device (PathTracerApp.cpp:60-72).
  void init(const OptionMap& args);
  /// Acquire the GPU(s) and upload constants: what GraphManager::run + build() do before execute().
  void attach();
  /// The host step loop (PathTracerApp.cpp:566-792).
  void execute();
  /// The 25 tool options of PathTracerApp.cpp:794-830 plus the 8 standard ones of main.cpp:8-37.
  static std::vector<OptionSpec> addToolOptions();

  double samplesPerSecond() const { return finalSamplesPerSec; }

private:
  bool loadNifModels(std::size_t numDevices, const std::string& assetPath);
  void initialiseState(std::uint32_t imageWidth, std::uint32_t imageHeight);

  OptionMap args;
  std::uint32_t samplesPerPixel = 0;
  std::uint32_t samplesPerIpuStep = 0;
  IpuJobList ipuJobs;
  DeviceGeometry geometry;
  std::vector<pt_handle> devices;
  std::vector<std::unique_ptr<NifModel>> models;
  std::unique_ptr<PathTracerState> traceState;
  double finalSamplesPerSec = 0.0;
};

std::size_t roundSamplesPerPixel(std::size_t samplesPerPixel, std::size_t samplesPerIpuStep);
