// extern "C" test shim over the host classes (for the Python test-suite; not part of the product boundary).
#include <cstdio>
#include <cstring>
#include <string>

#include "AccumulatedImage.hpp"
#include "AsyncTask.hpp"
#include <stdexcept>
#include "LoadBalancer.hpp"
#include "NifModel.hpp"
#include "PathTracerApp.hpp"
#include "image_io.hpp"

extern "C" {

std::size_t pth_calculate_max_rays_per_tile(std::size_t w, std::size_t h, std::size_t tiles, std::size_t workers) {
  DeviceGeometry g{tiles, workers};
  return calculateMaxRaysPerTile(w, h, g);
}

// Fills `out` (tiles * rays-per-tile records) with the shuffled, padded worklist; returns the record count.
std::size_t pth_make_shuffled_worklist(std::size_t w, std::size_t h, std::size_t tiles, std::size_t workers, TraceRecord* out,
                                       std::size_t capacity) {
  DeviceGeometry g{tiles, workers};
  auto jobs = createTracingJobs(w, h, g);
  LoadBalancer lb(jobs.size() * jobs.front().size());
  lb.randomiseWorkList(jobs);
  auto& list = lb.getWork().inactive();
  if (out && capacity >= list.size()) std::memcpy(out, list.data(), list.size() * sizeof(TraceRecord));
  return list.size();
}

// allocateWorkByPathLength + clearInactiveAccumulators on a caller-supplied list (in place); returns the path-length sum.
std::size_t pth_balance_and_clear(TraceRecord* records, std::size_t n, std::size_t jobs, int balance) {
  LoadBalancer lb(n);
  lb.getWork().inactive().assign(records, records + n);
  if (balance) {
    IpuJobList jl;
    OptionMap noArgs;
    for (std::size_t j = 0; j < jobs; ++j) jl.emplace_back(n / jobs, noArgs, j);
    lb.allocateWorkByPathLength(jl);
  }
  if (balance) std::memcpy(records, lb.getWork().inactive().data(), n * sizeof(TraceRecord));   // balanced order, lengths intact
  std::size_t sum = lb.clearInactiveAccumulators();
  if (!balance) std::memcpy(records, lb.getWork().inactive().data(), n * sizeof(TraceRecord));  // cleared accumulators
  return sum;
}

// dealTilesByPathLength: owner[t] for every tile; tileWorkList: the padded worklist of one device under that deal.
void pth_deal_tiles(const unsigned long long* cost, std::size_t n_tiles, std::size_t devices, int* owner_out) {
  std::vector<std::uint64_t> c(cost, cost + n_tiles);
  auto owner = dealTilesByPathLength(c, devices);
  for (std::size_t t = 0; t < n_tiles; ++t) owner_out[t] = owner[t];
}
long pth_tile_worklist(std::size_t w, std::size_t h, const int* owner, std::size_t n_tiles, int device, std::size_t pad_to,
                       TraceRecord* out) {
  try {
    std::vector<std::int32_t> o(owner, owner + n_tiles);
    auto list = tileWorkList(w, h, o, device, pad_to);
    std::memcpy(out, list.data(), list.size() * sizeof(TraceRecord));
    return (long)list.size();
  } catch (const std::exception&) {
    return -1;
  }
}
std::size_t pth_max_tile_items(std::size_t w, std::size_t h, std::size_t devices) { return maxTileItemsPerDevice(w, h, devices); }

// Accumulate `steps` identical record lists into a film, tone-map, save; copies the HDR (BGR) film out.
int pth_film_roundtrip(const TraceRecord* records, std::size_t n, std::size_t w, std::size_t h, std::size_t steps, float exposure,
                       float gamma, const char* file, float* hdr_out, unsigned char* ldr_out) {
  try {
    AccumulatedImage film(w, h);
    std::vector<TraceRecord> v(records, records + n);
    for (std::size_t s = 0; s < steps; ++s) film.accumulate(v);
    auto hdr = film.getHdrImage();
    std::memcpy(hdr_out, hdr.data.data(), hdr.data.size() * sizeof(float));
    const auto& ldr = film.updateLdrImage(steps, exposure, gamma);
    std::memcpy(ldr_out, ldr.data.data(), ldr.data.size());
    if (file && *file) film.saveImages(file, steps, exposure, gamma);
    return 0;
  } catch (...) { return -1; }
}

int pth_read_exr(const char* file, float* bgr, std::size_t capacity, std::size_t* w, std::size_t* h) {
  std::vector<float> d;
  if (!image_io::readExr(file, d, *w, *h)) return -1;
  if (d.size() > capacity) return -2;
  std::memcpy(bgr, d.data(), d.size() * sizeof(float));
  return 0;
}

// Parses nif_metadata.txt; out = {embedding, hidden, layers, logToneMap, max, mean[3] (eps folded)}.
int pth_read_metadata(const char* file, double* out8) {
  try {
    NifMetaData m(file);
    out8[0] = (double)m.embeddingDimension; out8[1] = (double)m.hiddenSize; out8[2] = (double)m.layerCount;
    out8[3] = m.logToneMap; out8[4] = m.max; out8[5] = m.mean[0]; out8[6] = m.mean[1]; out8[7] = m.mean[2];
    return 0;
  } catch (...) { return -1; }
}

std::size_t pth_round_samples(std::size_t spp, std::size_t per_step) { return roundSamplesPerPixel(spp, per_step); }

// splitTilePixelsOverWorkers (IpuPathTraceJob.cpp:30-52): writes `workers` (start, end) pairs.
void pth_split_pixels(std::size_t pixels, std::size_t workers, std::size_t* out) {
  auto v = splitTilePixelsOverWorkers(pixels, workers);
  for (std::size_t i = 0; i < v.size(); ++i) { out[2 * i] = v[i].first; out[2 * i + 1] = v[i].second; }
}

// IpuPathTraceJob(maxRayCount, args, core) + buildGraph with the reference's CLI defaults; out = {pixelCount, tile,
// begin.device, begin.firstRecord, begin.recordCount, end.firstRecord, n worker intervals, cfg.width, cfg.height,
// cfg.roulette_depth}, fout = {cfg.refractive_index, cfg.stop_prob}.
int pth_job_build(std::size_t rays, std::size_t core, std::size_t device, std::size_t first, std::size_t* out, float* fout) {
  try {
    OptionMap args;
    for (auto& s : PathTracerApp::addToolOptions()) if (!s.required) args.values[s.name] = s.defaultValue;
    IpuPathTraceJob job(rays, args, core);
    pt_config cfg{};
    DeviceGeometry g;
    job.buildGraph(cfg, device, first, g, args);
    out[0] = job.getPixelCount(); out[1] = job.getTile();
    out[2] = job.beginTraceJob().device; out[3] = job.beginTraceJob().firstRecord; out[4] = job.beginTraceJob().recordCount;
    out[5] = job.endTraceJob().firstRecord; out[6] = job.workerIntervals().size();
    out[7] = cfg.width; out[8] = cfg.height; out[9] = cfg.roulette_depth;
    fout[0] = cfg.refractive_index; fout[1] = cfg.stop_prob;
    return (int)(IpuPathTraceJob::numChannels * 10 + IpuPathTraceJob::numRayDirComponents);
  } catch (...) { return -1; }
}

// AsyncTask: a job that throws must surface at waitForCompletion(), not terminate the process.
int pth_async_task_rethrows() {
  AsyncTask t;
  t.run([] { throw std::runtime_error("boom"); });
  try { t.waitForCompletion(); } catch (const std::runtime_error& e) { return std::string(e.what()) == "boom" ? 1 : -1; }
  return 0;
}

}  // extern "C"

// ---- HDF5 reader shim (tests)
#include "Hdf5Reader.hpp"
extern "C" {
// Writes a '\n'-joined listing "name" per link of `group` into out; returns the count or -1 (message in out).
int pth_h5_list(const char* file, const char* group, char* out, std::size_t cap) {
  try {
    h5::File f(file);
    std::string s;
    auto names = f.listGroup(group);
    for (auto& n : names) s += n + "\n";
    std::snprintf(out, cap, "%s", s.c_str());
    return (int)names.size();
  } catch (const std::exception& e) { std::snprintf(out, cap, "%s", e.what()); return -1; }
}
int pth_h5_attr(const char* file, const char* object, const char* name, char* out, std::size_t cap) {
  try {
    h5::File f(file);
    std::string s = f.readStringAttribute(object, name);
    if (s.size() + 1 > cap) return -2;
    std::memcpy(out, s.data(), s.size());
    out[s.size()] = 0;
    return (int)s.size();
  } catch (const std::exception& e) { std::snprintf(out, cap, "%s", e.what()); return -1; }
}
// shape into dims[0..7] (rank returned), element size and raw bytes
int pth_h5_dataset(const char* file, const char* path, std::size_t* dims, std::size_t* elem, unsigned char* bytes, std::size_t cap,
                   char* err, std::size_t errcap) {
  try {
    h5::File f(file);
    h5::Dataset ds = f.openDataSet(path);
    for (std::size_t i = 0; i < ds.shape.size() && i < 8; ++i) dims[i] = ds.shape[i];
    *elem = ds.elementSize;
    if (bytes && cap >= ds.bytes.size()) std::memcpy(bytes, ds.bytes.data(), ds.bytes.size());
    return (int)ds.shape.size();
  } catch (const std::exception& e) { std::snprintf(err, errcap, "%s", e.what()); return -1; }
}
}

#include "Hdf5Model.hpp"
extern "C" {
// Loads a Keras H5 through Hdf5Model exactly as NifModel::Data::setupModel does; per layer writes
// {rows, cols, is_half, relu, has_bias, sum of the kernel's 16-bit words} into out; returns the layer count.
int pth_h5_model(const char* file, unsigned long long* out, std::size_t cap, char* err, std::size_t errcap) {
  try {
    Hdf5Model m(file);
    std::size_t i = 0;
    for (const auto& l : m.get()) {
      if (6 * (i + 1) > cap) break;
      unsigned long long sum = 0;
      const auto* w = reinterpret_cast<const std::uint16_t*>(l.kernelData.storage.data());
      for (std::size_t k = 0; k < l.kernelData.storage.size() / 2; ++k) sum += w[k];
      out[6 * i + 0] = l.kernelData.shape[0]; out[6 * i + 1] = l.kernelData.shape[1];
      out[6 * i + 2] = l.kernelData.isHalf(); out[6 * i + 3] = l.activation == "relu";
      out[6 * i + 4] = l.useBias; out[6 * i + 5] = sum;
      ++i;
    }
    return (int)i;
  } catch (const std::exception& e) { std::snprintf(err, errcap, "%s", e.what()); return -1; }
}
}
