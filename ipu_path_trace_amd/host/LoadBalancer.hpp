// Host-side worklist management with the reference's names and semantics (src/LoadBalancer.hpp, LoadBalancer.cpp):
// padded per-job worklists, the seeded global shuffle, path-length balancing and the accumulator reset that also
// returns the ray count.  poplar::Target is replaced by DeviceGeometry (IpuPathTraceJob.hpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

#include "IpuPathTraceJob.hpp"
#include "TraceRecord.hpp"

using RecordList = std::vector<TraceRecord>;

/// Rays every job traces for one sample per pixel: ceil(pixels / tiles), then `+= r % workers` (the reference's
/// rounding, LoadBalancer.cpp:27-35 -- deliberately not a round-up to a multiple), at least `workers`.
std::size_t calculateMaxRaysPerTile(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target);

/// One work item per pixel in row-major order (LoadBalancer.cpp:38-52).
RecordList createWorkListForImage(std::size_t imageWidth, std::size_t imageHeight);

/// Equal slices of the image worklist, one per job; the tail is padded with (65535, 65535) items that the film skips
/// (LoadBalancer.cpp:54-86).
std::vector<RecordList> createTracingJobs(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target);

/// Two equally sized record lists: the device owns `active()`, the host thread works on `inactive()`.
class WorkList {
public:
  explicit WorkList(std::size_t size) : lists{RecordList(size), RecordList(size)} {}
  virtual ~WorkList() = default;

  RecordList& active() { return lists[0]; }
  RecordList& inactive() { return lists[1]; }

  /// Exchange the roles; an empty new active list is a logic error (LoadBalancer.cpp:103-108).
  void swap() {
    std::swap(lists[0], lists[1]);
    if (lists[0].empty()) throw std::logic_error("The new active worklist is empty.");
  }

private:
  RecordList lists[2];
};

// ---- path-length balancing ACROSS devices, on image tiles (the unit of re-dealing between GPUs).  The reference pairs
// the shortest with the longest path inside every IPU tile (allocateWorkByPathLength, LoadBalancer.cpp:141-192); the
// same idea for D devices: sort the image tiles by measured cost and deal them in boustrophedon order (0..D-1, D-1..0,
// ...), which for two tiles per device IS the shortest+longest pairing.  Deterministic (stable sort, ties by tile id).
constexpr std::size_t kBalanceTile = 16;
inline std::size_t balanceTileCount(std::size_t w, std::size_t h) {
  return ((w + kBalanceTile - 1) / kBalanceTile) * ((h + kBalanceTile - 1) / kBalanceTile);
}
/// owner[t] = device of tile t.
std::vector<std::int32_t> dealTilesByPathLength(const std::vector<std::uint64_t>& cost, std::size_t devices);
/// Work items of `device` under `owner`: its tiles in tile order, pixels row-major inside a tile, padded to `padTo`.
RecordList tileWorkList(std::size_t imageWidth, std::size_t imageHeight, const std::vector<std::int32_t>& owner,
                        std::int32_t device, std::size_t padTo);
/// Capacity that fits any deal with equal tile counts (+-1).
inline std::size_t maxTileItemsPerDevice(std::size_t w, std::size_t h, std::size_t devices) {
  return (balanceTileCount(w, h) + devices - 1) / devices * kBalanceTile * kBalanceTile;
}

class LoadBalancer {
public:
  explicit LoadBalancer(std::size_t workItemCount) : work(workItemCount) {}
  virtual ~LoadBalancer() = default;

  WorkList& getWork() { return work; }

  /// Flatten the jobs, shuffle with std::mt19937(142) and store as the inactive list (LoadBalancer.cpp:118-139).
  void randomiseWorkList(const std::vector<RecordList>& jobs);
  /// Re-deal the inactive list so every job gets shortest+longest path pairs (LoadBalancer.cpp:141-192).
  void allocateWorkByPathLength(const IpuJobList& jobs);
  /// Zero r,g,b,sampleCount,pathLength of the inactive list; returns the summed path lengths (LoadBalancer.cpp:198-213).
  std::size_t clearInactiveAccumulators();
  void clearActiveAccumulators();

private:
  WorkList work;
};
