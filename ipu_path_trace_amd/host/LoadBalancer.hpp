// Host worklist management: reference src/LoadBalancer.hpp:14-56 (same free functions and classes).
#pragma once
#include <cstddef>
#include <vector>

#include "IpuPathTraceJob.hpp"
#include "TraceRecord.hpp"

using RecordList = std::vector<TraceRecord>;

/// LoadBalancer.cpp:14-36 (including its `r += r % workers` rounding, which is not a true round-up).
std::size_t calculateMaxRaysPerTile(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target);
/// LoadBalancer.cpp:38-52: one item per pixel, row-major (c, r).
std::vector<TraceRecord> createWorkListForImage(std::size_t imageWidth, std::size_t imageHeight);
/// LoadBalancer.cpp:54-86: pad with (65535, 65535) items and cut into equal per-tile lists.
std::vector<RecordList> createTracingJobs(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target);

/// A double buffered work list (LoadBalancer.cpp:88-108).
struct WorkList {
  WorkList(std::size_t size);
  virtual ~WorkList();
  void swap();
  RecordList& active();
  RecordList& inactive();

private:
  RecordList activeWork;
  RecordList inactiveWork;
};

struct LoadBalancer {
  LoadBalancer(std::size_t workItemCount);
  virtual ~LoadBalancer();

  WorkList& getWork() { return work; }

  void randomiseWorkList(const std::vector<RecordList>& jobs);   // LoadBalancer.cpp:118-139
  void allocateWorkByPathLength(const IpuJobList& jobs);         // LoadBalancer.cpp:141-192
  std::size_t clearInactiveAccumulators();                       // LoadBalancer.cpp:198-213
  void clearActiveAccumulators();                                // LoadBalancer.cpp:216-225

private:
  WorkList work;
};
