// Per-"tile" job descriptor: reference src/IpuPathTraceJob.hpp:32-93.
//
// On the IPU a job wires the five codelets onto one tile.  On MI355X the device program is a fixed set
// of HIP kernels behind include/ptmi.h, so a job keeps only what the host side consumes: the size of its
// slice of the worklist and its ordinal.  The public members the LoadBalancer and PathTracerApp use
// (constructor shape, getPixelCount, getTile, numChannels, numRayDirComponents) are unchanged.
#pragma once
#include <cstddef>
#include <vector>

struct IpuPathTraceJob;
using IpuJobList = std::vector<IpuPathTraceJob>;

/// What LoadBalancer.cpp reads from poplar::Target: tiles and worker contexts (LoadBalancer.cpp:15-16).
/// Defaults are one Mk2 IPU (1472 tiles x 6 workers) so worklist shapes match the reference exactly.
struct DeviceGeometry {
  std::size_t numTiles = 1472;
  std::size_t numWorkerContexts = 6;
  std::size_t getNumTiles() const { return numTiles; }
  std::size_t getNumWorkerContexts() const { return numWorkerContexts; }
};

struct IpuPathTraceJob {
  std::size_t maxPixelCount;

  IpuPathTraceJob(std::size_t maxRayCount, std::size_t core) : maxPixelCount(maxRayCount), ipuCore(core) {}

  std::size_t getPixelCount() const { return maxPixelCount; }
  std::size_t getTile() const { return ipuCore; }

  static constexpr std::size_t numChannels = 3;
  static constexpr std::size_t numRayDirComponents = 2;

private:
  std::size_t ipuCore;
};
