// Per-"tile" job: reference src/IpuPathTraceJob.hpp:32-93, IpuPathTraceJob.cpp:30-214.
//
// On the IPU a job wires the five codelets onto one tile (buildGraph) and contributes two program fragments
// (beginTraceJob / endTraceJob) to the iteration program.  On MI355X the device program is a fixed set of HIP kernels
// behind include/ptmi.h, so a job has no vertices to add; what it still owns is exactly what the reference's job owns
// on the host side: its slice of the device's trace buffer, the split of that slice over worker contexts
// (splitTilePixelsOverWorkers, IpuPathTraceJob.cpp:30-52) and the per-tile scalars buildGraph connects
// (imageWidth/Height, refractiveIndex, rouletteDepth, stopProb: IpuPathTraceJob.cpp:95-138) -- here they are written
// into the owning device's pt_config.  Interface kept: ctor (maxRayCount, args, core), buildGraph, beginTraceJob,
// endTraceJob, getPixelCount, getTile, numChannels, numRayDirComponents.
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

#include "Options.hpp"
#include "ptmi.h"

struct IpuPathTraceJob;
using IpuJobList = std::vector<IpuPathTraceJob>;
using Interval = std::pair<std::size_t, std::size_t>;

/// Start and end indices that slice a tile's pixels into the chunks each worker processes: equal shares, leftovers to
/// the first workers (IpuPathTraceJob.cpp:30-52).
std::vector<Interval> splitTilePixelsOverWorkers(std::size_t pixelCount, std::size_t workers);

/// What LoadBalancer.cpp reads from poplar::Target: tiles and worker contexts (LoadBalancer.cpp:15-16).
/// Defaults are one Mk2 IPU (1472 tiles x 6 workers) so worklist shapes match the reference exactly.
struct DeviceGeometry {
  std::size_t numTiles = 1472;
  std::size_t numWorkerContexts = 6;
  std::size_t getNumTiles() const { return numTiles; }
  std::size_t getNumWorkerContexts() const { return numWorkerContexts; }
};

/// The role poplar::program::Sequence plays for beginTraceJob()/endTraceJob(): the part of a device's launch a job
/// stands for.  `begin` marks the records the job's rays are generated from, `end` the records its results land in
/// (the same slice: TraceRecord is read and accumulated in place, codelets.cpp:47-79,241-304).
struct TraceJobProgram {
  std::size_t device = 0;        // device ordinal whose trace buffer holds the slice
  std::size_t firstRecord = 0;   // offset into that device's trace buffer
  std::size_t recordCount = 0;
};

struct IpuPathTraceJob {
  std::size_t maxPixelCount;

  ~IpuPathTraceJob();

  /// Only initialises values independent of device set-up, as in the reference (`args` is accepted and unused there
  /// too: IpuPathTraceJob.cpp:72-76).
  IpuPathTraceJob(std::size_t maxRayCount, const OptionMap& args, std::size_t core);

  /// The host half of the reference's buildGraph: place the job's slice in its device's trace buffer, split it over
  /// `target`'s worker contexts and write the per-tile scalars the reference connects to its vertices into `config`
  /// (every job of a device writes the same values, as every tile gets the same constants).
  void buildGraph(pt_config& config, std::size_t device, std::size_t firstRecord, const DeviceGeometry& target,
                  const OptionMap& args);

  TraceJobProgram beginTraceJob() const { return beginSeq; }
  TraceJobProgram endTraceJob() const { return endSeq; }

  std::size_t getPixelCount() const { return maxPixelCount; }
  std::size_t getTile() const { return ipuCore; }
  const std::vector<Interval>& workerIntervals() const { return intervals; }

  static constexpr std::size_t numChannels = 3;
  static constexpr std::size_t numRayDirComponents = 2;

private:
  std::size_t ipuCore;  // Core instead of 'Tile' to avoid confusion with the image tiles.

  // assigned by buildGraph only
  std::vector<Interval> intervals;
  TraceJobProgram beginSeq;
  TraceJobProgram endSeq;
};
