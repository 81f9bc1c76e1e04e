// Application object: reference src/PathTracerApp.hpp:36-57.  init() / execute() / addToolOptions() keep
// their roles; build() (Poplar graph construction) has no counterpart -- the device program is libptmi.so.
#pragma once
#include <chrono>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "AccumulatedImage.hpp"
#include "InterfaceServer.hpp"
#include "IpuPathTraceJob.hpp"
#include "LoadBalancer.hpp"
#include "NifModel.hpp"
#include "Options.hpp"
#include "ptmi.h"

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
  /// One call on every device, each from its own thread (devices run concurrently and exchange no ray data,
  /// PathTracerApp.cpp:205-252); throws with the first device's message on failure.
  template <class F> void onEveryDevice(const char* what, F&& call);
  /// Step loop with the film resident on the devices: path_trace + pt_film_accumulate per step, one RCCL gather of HDR
  /// tiles to device 0 at every save interval.
  void executeResidentFilm(std::uint32_t steps);
  /// User interaction invalidates all rendering in progress: new (or recycled) tracer state is swapped in and the up-to-date
  /// worklist copied over, so nobody waits for the defunct host task (PathTracerApp.cpp:507-528).
  void defunctState(std::uint32_t imageWidth, std::uint32_t imageHeight);
  /// PathTracerApp.cpp:530-564: stop / detach / NIF hot-reload + restart.
  InterfaceServer::Status processUserInput(InterfaceServer::State& state, std::uint32_t imageWidth, std::uint32_t imageHeight);
  /// Step loop as the reference runs it (setup -> path_trace -> read_results, host film): needed when the balancer
  /// re-deals the worklist from the returned path lengths every step.
  void executeHostFilm(std::uint32_t steps);

  OptionMap args;
  std::uint32_t samplesPerPixel = 0;
  std::uint32_t samplesPerIpuStep = 0;
  IpuJobList ipuJobs;
  DeviceGeometry geometry;
  std::vector<pt_handle> devices;
  std::vector<std::unique_ptr<NifModel>> models;
  std::unique_ptr<PathTracerState> traceState;
  std::unique_ptr<PathTracerState> defunctTraceState;   // keeps defunct data alive while the async host task finishes on it
  std::size_t deviceCapacity = 0;   // work items per device incl. padding (pt_config.max_work_items)
  bool hostGather = false;   ///< HDR tiles through the host (one copy per device) instead of the RCCL gather
  double finalSamplesPerSec = 0.0;
  std::chrono::steady_clock::time_point renderStartTime;   // reset when the UI restarts the render (PathTracerApp.cpp:669)
};

std::size_t roundSamplesPerPixel(std::size_t samplesPerPixel, std::size_t samplesPerIpuStep);
