// Reference: src/LoadBalancer.cpp.  Behaviour kept, including the quirks listed in SURVEY.md's appendix.
#include "LoadBalancer.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <random>
#include <stdexcept>

#include "logging.hpp"

std::size_t calculateMaxRaysPerTile(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target) {
  const auto numTiles = target.getNumTiles();
  const auto numWorkers = target.getNumWorkerContexts();
  if ((imageWidth * imageHeight) % (numTiles * numWorkers)) {
    pt_log::warn_("For best performance number of pixels in image should be divisible by {} x {} (tiles x workers).",
                  numTiles, numWorkers);
  }
  const auto totalRayCount = imageWidth * imageHeight;
  unsigned raysPerTile = std::ceil(totalRayCount / (float)numTiles);
  raysPerTile += raysPerTile % numWorkers;  // sic: reference LoadBalancer.cpp:31
  return std::max<std::size_t>(numWorkers, raysPerTile);
}

std::vector<TraceRecord> createWorkListForImage(std::size_t imageWidth, std::size_t imageHeight) {
  std::vector<TraceRecord> workList;
  workList.reserve(imageWidth * imageHeight);
  for (std::size_t r = 0; r < imageHeight; ++r)
    for (std::size_t c = 0; c < imageWidth; ++c) workList.emplace_back(c, r);
  return workList;
}

std::vector<RecordList> createTracingJobs(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target) {
  const auto numTiles = target.getNumTiles();
  const auto maxRaysPerTile = calculateMaxRaysPerTile(imageWidth, imageHeight, target);
  auto paddedRayCount = maxRaysPerTile * numTiles;
  auto workList = createWorkListForImage(imageWidth, imageHeight);
  const auto dummyCoord = std::numeric_limits<std::uint16_t>::max();
  while (workList.size() < paddedRayCount) workList.emplace_back(dummyCoord, dummyCoord);
  auto copyItr = workList.cbegin();
  std::vector<RecordList> perTileWork;
  perTileWork.reserve(numTiles);
  for (std::size_t t = 0; t < numTiles; ++t) {
    perTileWork.emplace_back(maxRaysPerTile);
    auto endItr = copyItr + maxRaysPerTile;
    std::copy(copyItr, endItr, perTileWork.back().begin());
    copyItr = endItr;
  }
  return perTileWork;
}

WorkList::WorkList(std::size_t size) : activeWork(size), inactiveWork(size) {}
WorkList::~WorkList() {}
RecordList& WorkList::active() { return activeWork; }
RecordList& WorkList::inactive() { return inactiveWork; }

void WorkList::swap() {
  std::swap(activeWork, inactiveWork);
  if (activeWork.empty()) throw std::logic_error("The new active worklist is empty.");
}

LoadBalancer::LoadBalancer(std::size_t workItemCount) : work(workItemCount) {}
LoadBalancer::~LoadBalancer() {}

void LoadBalancer::randomiseWorkList(const std::vector<RecordList>& jobs) {
  std::vector<TraceRecord> workList;
  workList.reserve(jobs.size() * jobs.front().size());
  for (const auto& j : jobs)
    for (const auto& w : j) workList.push_back(w);
  auto workSeed = 142u;
  std::mt19937 g(workSeed);
  std::shuffle(workList.begin(), workList.end(), g);
  work.inactive() = workList;
}

void LoadBalancer::allocateWorkByPathLength(const IpuJobList& jobs) {
  auto sorted = work.inactive();
  std::sort(sorted.begin(), sorted.end(),
            [](const TraceRecord& a, const TraceRecord& b) -> bool { return a.pathLength < b.pathLength; });
  std::vector<RecordList> perTileWork(jobs.size());
  for (auto& t : perTileWork) t.reserve(jobs[0].getPixelCount());
  auto shortItr = sorted.begin();
  auto longItr = sorted.end() - 1;
  pt_log::info_("Load balancing started ({} work items)", sorted.size());
  pt_log::info_("Path length min/max: {}/{}", shortItr->pathLength, longItr->pathLength);
  // Each job takes the shortest and the longest remaining path in turn (reference :168-181; like the
  // reference this assumes an even item count per job).
  while (true) {
    for (auto& t : perTileWork) {
      t.push_back(*shortItr);
      t.push_back(*longItr);
      ++shortItr;
      --longItr;
    }
    if (longItr <= shortItr) break;
  }
  pt_log::info_("Load balancing finished");
  auto itr = sorted.begin();
  for (auto& t : perTileWork)
    for (auto& w : t) {
      if (itr == sorted.end()) break;
      *itr = w;
      ++itr;
    }
  work.inactive() = sorted;
}

std::size_t LoadBalancer::clearInactiveAccumulators() {
  auto& list = work.inactive();
  std::size_t sum = 0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
  for (std::size_t i = 0; i < list.size(); ++i) {
    auto& t = list[i];
    sum += t.pathLength;
    t.r = t.g = t.b = 0.f;
    t.pathLength = 0;
    t.sampleCount = 0;
  }
  return sum;
}

void LoadBalancer::clearActiveAccumulators() {
  auto& list = work.active();
#pragma omp parallel for schedule(static)
  for (std::size_t i = 0; i < list.size(); ++i) {
    auto& t = list[i];
    t.r = t.g = t.b = 0.f;
    t.pathLength = 0;
    t.sampleCount = 0;
  }
}
