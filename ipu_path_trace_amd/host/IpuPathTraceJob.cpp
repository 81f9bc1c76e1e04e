#include "IpuPathTraceJob.hpp"

#include "logging.hpp"

std::vector<Interval> splitTilePixelsOverWorkers(std::size_t pixelCount, std::size_t workers) {
  std::vector<Interval> out;
  if (workers == 0) return out;
  out.reserve(workers);
  const std::size_t share = pixelCount / workers, extra = pixelCount % workers;
  pt_log::trace_("Worker split: total rays: {} rays per-worker: {} leftovers: {}", pixelCount, share, extra);
  std::size_t at = 0;
  for (std::size_t w = 0; w < workers; ++w) {
    const std::size_t n = share + (w < extra ? 1 : 0);
    out.emplace_back(at, at + n);
    at += n;
  }
  return out;
}

IpuPathTraceJob::~IpuPathTraceJob() {}

IpuPathTraceJob::IpuPathTraceJob(std::size_t maxRayCount, const OptionMap&, std::size_t core)
    : maxPixelCount(maxRayCount), ipuCore(core) {}

void IpuPathTraceJob::buildGraph(pt_config& config, std::size_t device, std::size_t firstRecord, const DeviceGeometry& target,
                                 const OptionMap& args) {
  config.width = args.u32("width");                         // IpuPathTraceJob.cpp:95-98
  config.height = args.u32("height");
  config.refractive_index = args.f32("refractive-index");   // :133 (rounded to half on the device, as the HALF constant is)
  config.roulette_depth = args.u32("roulette-depth");       // :135
  config.stop_prob = args.f32("stop-prob");                 // :137
  intervals = splitTilePixelsOverWorkers(getPixelCount(), target.getNumWorkerContexts());
  beginSeq = TraceJobProgram{device, firstRecord, getPixelCount()};
  endSeq = beginSeq;
}
