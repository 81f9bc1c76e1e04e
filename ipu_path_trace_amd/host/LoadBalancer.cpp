#include "host_threads.hpp"
#include "LoadBalancer.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <random>
#include <stdexcept>

#include "logging.hpp"

namespace {

constexpr std::uint16_t kPaddingCoord = std::numeric_limits<std::uint16_t>::max();
constexpr unsigned kShuffleSeed = 142u;   // LoadBalancer.cpp:133

std::size_t resetAndSumPathLengths(RecordList& list) {
  std::size_t total = 0;
  const std::size_t n = list.size();
#pragma omp parallel for reduction(+ : total) schedule(static) num_threads(hostLoopThreads())
  for (std::size_t i = 0; i < n; ++i) {
    total += list[i].pathLength;
    list[i].clearAccumulators();
  }
  return total;
}

}  // namespace

std::size_t calculateMaxRaysPerTile(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target) {
  const std::size_t tiles = target.getNumTiles(), workers = target.getNumWorkerContexts();
  const std::size_t pixels = imageWidth * imageHeight;
  if (pixels % (tiles * workers) != 0) {
    pt_log::warn_("For best performance number of pixels in image should be divisible by {} x {} (tiles x workers).",
                  tiles, workers);
  }
  unsigned rays = static_cast<unsigned>(std::ceil(pixels / static_cast<float>(tiles)));   // float division as the reference
  rays += rays % workers;
  return std::max<std::size_t>(workers, rays);
}

RecordList createWorkListForImage(std::size_t imageWidth, std::size_t imageHeight) {
  RecordList items(imageWidth * imageHeight);
  for (std::size_t i = 0; i < items.size(); ++i)
    items[i] = TraceRecord(static_cast<std::uint16_t>(i % imageWidth), static_cast<std::uint16_t>(i / imageWidth));
  return items;
}

std::vector<RecordList> createTracingJobs(std::size_t imageWidth, std::size_t imageHeight, const DeviceGeometry& target) {
  const std::size_t perJob = calculateMaxRaysPerTile(imageWidth, imageHeight, target);
  const RecordList pixels = createWorkListForImage(imageWidth, imageHeight);
  const TraceRecord padding(kPaddingCoord, kPaddingCoord);
  std::vector<RecordList> jobs(target.getNumTiles(), RecordList(perJob, padding));
  for (std::size_t i = 0; i < pixels.size(); ++i) jobs[i / perJob][i % perJob] = pixels[i];
  return jobs;
}

void LoadBalancer::randomiseWorkList(const std::vector<RecordList>& jobs) {
  RecordList flat;
  flat.reserve(jobs.size() * jobs.front().size());
  for (const RecordList& job : jobs) flat.insert(flat.end(), job.begin(), job.end());
  std::mt19937 generator(kShuffleSeed);
  std::shuffle(flat.begin(), flat.end(), generator);
  work.inactive() = std::move(flat);
}

void LoadBalancer::allocateWorkByPathLength(const IpuJobList& jobs) {
  RecordList& list = work.inactive();
  RecordList byLength = list;
  std::sort(byLength.begin(), byLength.end(),
            [](const TraceRecord& a, const TraceRecord& b) { return a.pathLength < b.pathLength; });
  pt_log::info_("Load balancing started ({} work items)", byLength.size());
  pt_log::info_("Path length min/max: {}/{}", byLength.front().pathLength, byLength.back().pathLength);

  // Deal (shortest, longest) pairs to the jobs in turn until the two ends of the sorted list meet.  The reference
  // (LoadBalancer.cpp:160-177) tests `hi > lo` only after a whole round over the jobs, which duplicates and drops items
  // (and reads out of range) when the item count is not a multiple of 2 x jobs; it never is with the default geometry
  // (6 workers force an even count per job).  Here the ends are tested before every deal and a single middle item is
  // dealt once, so the result is a permutation of the input for any count -- identical to the reference's where the
  // reference's is defined.
  const std::size_t nJobs = jobs.size();
  if (nJobs == 0) throw std::logic_error("allocateWorkByPathLength needs at least one job.");
  std::vector<RecordList> dealt(nJobs);
  for (RecordList& d : dealt) d.reserve(jobs.front().getPixelCount());
  std::ptrdiff_t lo = 0, hi = static_cast<std::ptrdiff_t>(byLength.size()) - 1;
  while (lo <= hi) {
    for (std::size_t j = 0; j < nJobs && lo <= hi; ++j) {
      dealt[j].push_back(byLength[static_cast<std::size_t>(lo++)]);
      if (lo <= hi) dealt[j].push_back(byLength[static_cast<std::size_t>(hi--)]);
    }
  }
  pt_log::info_("Load balancing finished");

  std::size_t out = 0;
  for (const RecordList& d : dealt)
    for (const TraceRecord& item : d)
      if (out < list.size()) list[out++] = item;
}

std::vector<std::int32_t> dealTilesByPathLength(const std::vector<std::uint64_t>& cost, std::size_t devices) {
  if (devices == 0) throw std::logic_error("dealTilesByPathLength needs at least one device.");
  std::vector<std::size_t> order(cost.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return cost[a] > cost[b]; });
  std::vector<std::int32_t> owner(cost.size());
  for (std::size_t pos = 0; pos < order.size(); ++pos) {
    const std::size_t round = pos / devices, idx = pos % devices;
    owner[order[pos]] = static_cast<std::int32_t>(round % 2 == 0 ? idx : devices - 1 - idx);
  }
  return owner;
}

RecordList tileWorkList(std::size_t imageWidth, std::size_t imageHeight, const std::vector<std::int32_t>& owner,
                        std::int32_t device, std::size_t padTo) {
  const std::size_t tx = (imageWidth + kBalanceTile - 1) / kBalanceTile;
  RecordList items;
  items.reserve(padTo);
  for (std::size_t t = 0; t < owner.size(); ++t) {
    if (owner[t] != device) continue;
    const std::size_t r0 = (t / tx) * kBalanceTile, c0 = (t % tx) * kBalanceTile;
    for (std::size_t dy = 0; dy < kBalanceTile; ++dy)
      for (std::size_t dx = 0; dx < kBalanceTile; ++dx)
        if (r0 + dy < imageHeight && c0 + dx < imageWidth)
          items.emplace_back(static_cast<std::uint16_t>(c0 + dx), static_cast<std::uint16_t>(r0 + dy));
  }
  if (items.size() > padTo) throw std::logic_error("tileWorkList: the deal does not fit the device's capacity.");
  items.resize(padTo, TraceRecord(kPaddingCoord, kPaddingCoord));
  return items;
}

std::size_t LoadBalancer::clearInactiveAccumulators() { return resetAndSumPathLengths(work.inactive()); }

void LoadBalancer::clearActiveAccumulators() { (void)resetAndSumPathLengths(work.active()); }
