#include "PathTracerApp.hpp"
#include "image_io.hpp"

#include <sstream>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <fstream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <thread>

#include "AsyncTask.hpp"
#include "logging.hpp"
#include "trace_ranges.hpp"

/// Adjust samples per pixel to be a multiple of samples per step (PathTracerApp.cpp:19-27).
std::size_t roundSamplesPerPixel(std::size_t samplesPerPixel, std::size_t samplesPerIpuStep) {
  if (samplesPerPixel % samplesPerIpuStep) {
    samplesPerPixel += samplesPerIpuStep - (samplesPerPixel % samplesPerIpuStep);
    pt_log::info_("Rounding SPP to next multiple of {}  (Rounded SPP :=  {})", samplesPerIpuStep, samplesPerPixel);
  }
  return samplesPerPixel;
}

namespace {
void check(pt_handle h, int rc, const char* what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + pt_last_error(h));
}
int aaNoiseType(const std::string& s) {
  if (s == "normal") return PT_AA_NORMAL;
  if (s == "uniform") return PT_AA_UNIFORM;
  if (s == "truncated-normal") return PT_AA_TRUNCATED_NORMAL;
  throw std::runtime_error("Invalid AA noise type: " + s);  // PathTracerApp.cpp:43
}
}  // namespace

PathTracerApp::PathTracerApp() {}

PathTracerApp::~PathTracerApp() {
  for (auto h : devices) pt_destroy(h);
}

std::vector<OptionSpec> PathTracerApp::addToolOptions() {
  return {
      // main.cpp:8-37
      {"help", 0, "", false, true, "Show command help."},
      {"model", 0, "false", false, true, "IPU-only (simulator): accepted and ignored."},
      {"ipus", 0, "1", false, false, "Number of devices to use (MI355X GPUs here)."},
      {"save-exe", 0, "", false, false, "IPU-only (graph cache): accepted and ignored."},
      {"load-exe", 0, "", false, false, "IPU-only (graph cache): accepted and ignored."},
      {"compile-only", 0, "false", false, true, "Validate the options and the NIF assets, then stop without touching a device (the reference compiles its graph and stops)."},
      {"defer-attach", 0, "false", false, true, "IPU-only: accepted and ignored."},
      {"log-level", 0, "info", false, false, "One of 'trace', 'debug', 'info', 'warn', 'err', 'critical', 'off'."},
      // PathTracerApp.cpp:794-830
      {"outfile", 'o', "", true, false, "Set output file name."},
      {"save-interval", 0, "1", false, false, ""},
      {"width", 'w', "256", false, false, "Output image width (total pixels)."},
      {"height", 'h', "256", false, false, "Output image height (total pixels)."},
      {"samples", 's', "512", false, false, "Total samples to take per pixel."},
      {"samples-per-step", 0, "512", false, false, "Samples to take per device step."},
      {"interactive-samples", 0, "8", false, false, "Number of samples to take per step during user interaction."},
      {"refractive-index", 'n', "1.5", false, false, "Refractive index."},
      {"roulette-depth", 0, "3", false, false, "Number of bounces before rays are randomly stopped."},
      {"stop-prob", 0, "0.3", false, false, "Probability of a ray being stopped."},
      {"aa-noise-scale", 'a', "0.3", false, false, "Scale of anti-aliasing noise (pixels)."},
      {"fov", 0, "90", false, false, "Horizontal field of view (degrees)."},
      {"exposure", 0, "0", false, false, "Exposure compensation for tone-mapping."},
      {"gamma", 0, "2.2", false, false, "Gamma correction for tone-mapping."},
      {"env-map-rotation", 0, "0", false, false, "Azimuthal rotation for HDRI environment map (degrees)."},
      {"seed", 0, "1", false, false, "Seed for random number generation."},
      {"aa-noise-type", 0, "normal", false, false, "['uniform', 'normal', 'truncated-normal']."},
      {"codelet-path", 0, "./", false, false, "IPU-only: accepted and ignored."},
      {"enable-load-balancing", 0, "false", false, true, "Run dynamic load balancing algorithm for path tracing."},
      {"max-path-length", 0, "10", false, false, ""},
      {"assets", 0, "", true, false, "Path to the 'assets.extra' directory of the saved keras model."},
      {"partials-type", 0, "half", false, false, "IPU-only: accepted and ignored (MFMA accumulates in fp32)."},
      {"available-memory-proportion", 0, "0.6", false, false, "IPU-only: accepted and ignored."},
      {"max-nif-batch-size", 0, "44160", false, false, "IPU-only: accepted and ignored (the NIF runs on a compacted queue)."},
      {"ui-port", 0, "0", false, false, "Start a remote user-interface server on the specified port (text protocol, see InterfaceServer.hpp)."},
      // additions of this build
      {"synthetic-nif", 0, "false", false, true, "Use seeded stand-in NIF weights when <assets>/converted.ptnif is absent."},
      {"constant-env", 0, "", false, false, "r,g,b: constant-radiance environment instead of the NIF (BASELINE config C1)."},
      {"host-film", 0, "false", false, true, "Run the reference's step loop (worklist to the host every step, host film) even without load balancing."},
      {"devices", 0, "", false, false, "GPU ordinal of every logical device, e.g. 0,1,2,3 (default: 0 .. ipus-1). Several logical devices may share a GPU (0,0): HDR tiles are then gathered through the host."},
      {"host-gather", 0, "false", false, true, "Gather the HDR tiles of the devices through the host (one copy per device) instead of over an RCCL communicator."},
  };
}

void PathTracerApp::init(const OptionMap& options) {
  args = options;
  samplesPerPixel = args.u32("samples");
  samplesPerIpuStep = args.u32("samples-per-step");
  if (samplesPerIpuStep) samplesPerPixel = (std::uint32_t)roundSamplesPerPixel(samplesPerPixel, samplesPerIpuStep);
  // values the reference would divide by (PathTracerApp.cpp:232 `step % saveInterval`, tiles / ipus): reject them up front
  if (args.u32("ipus") == 0) throw std::runtime_error("--ipus must be at least 1.");
  if (args.u32("save-interval") == 0) throw std::runtime_error("--save-interval must be at least 1.");
  if (samplesPerIpuStep == 0) throw std::runtime_error("--samples-per-step must be at least 1.");
  // the reference hands --outfile to cv::imwrite, which picks the codec by extension (AccumulatedImage.cpp:49) and throws for one
  // it has no writer for -- here before anything is rendered, not at the first save interval
  if (!image_io::ldrWriterFor(args.str("outfile")))
    throw std::runtime_error("--outfile '" + args.str("outfile") + "': could not find a writer for the specified extension "
                             "(built in: .png .jpg .bmp .ppm .pnm .tif .tiff; the HDR image goes to <name>.exr)");
  if (!args.has("constant-env") || args.str("constant-env").empty()) {
    if (!loadNifModels(args.u32("ipus"), args.str("assets"))) throw std::runtime_error("Could not load NIF model.");
  }
}

bool PathTracerApp::loadNifModels(std::size_t numDevices, const std::string& assetPath) {
  try {
    const auto metaFile = assetPath + "/nif_metadata.txt";
    const auto h5File = assetPath + "/converted.hdf5";          // the reference's asset (PathTracerApp.cpp:110)
    const auto weightFile = assetPath + "/converted.ptnif";      // flat side-car written by nif_assets.write_ptnif
    std::shared_ptr<NifModel::Data> nifData;
    if (std::ifstream(h5File).good()) {
      nifData = std::make_shared<NifModel::Data>(h5File, metaFile);
    } else if (std::ifstream(weightFile).good()) {
      nifData = std::make_shared<NifModel::Data>(weightFile, metaFile);
    } else if (args.flag("synthetic-nif")) {
      pt_log::warn_("'{}' not found: using seeded synthetic NIF weights", weightFile);
      nifData = NifModel::Data::synthetic(metaFile, 2024u);
    } else {
      throw std::runtime_error("neither '" + h5File + "' nor '" + weightFile + "' found (pass --synthetic-nif for stand-in weights)");
    }
    models.clear();
    for (std::size_t c = 0; c < numDevices; ++c)
      models.push_back(std::make_unique<NifModel>(nifData, "env_nif_gpu" + std::to_string(c)));
  } catch (std::exception& e) {
    pt_log::error_("Could not load NIF model from '{}'. Exception: {}", assetPath, e.what());
    return false;
  }
  return true;
}

void PathTracerApp::attach() {
  if (pt_abi_version() != PTMI_ABI_VERSION)   // built against one include/ptmi.h, running on another libptmi.so
    throw std::runtime_error("libptmi.so has ABI version " + std::to_string(pt_abi_version()) + ", this host was built against " +
                             std::to_string(PTMI_ABI_VERSION));
  const auto imageWidth = args.u32("width"), imageHeight = args.u32("height");
  const std::size_t numDevices = args.u32("ipus");
  geometry.numTiles *= numDevices;  // tiles scale with the device count as on a multi-IPU target
  // logical device d runs on GPU deviceMap[d] (default d).  RCCL needs one GPU per rank: logical devices that share a GPU
  // (a one-GPU box rehearsing the multi-device loop) gather their HDR tiles through the host instead.
  std::vector<std::int32_t> deviceMap(numDevices);
  for (std::size_t d = 0; d < numDevices; ++d) deviceMap[d] = (std::int32_t)d;
  if (args.has("devices") && !args.str("devices").empty()) {
    std::vector<std::int32_t> listed;
    std::stringstream list(args.str("devices"));
    for (std::string item; std::getline(list, item, ',');) {
      std::size_t used = 0;
      int v = -1;
      try { v = std::stoi(item, &used); } catch (const std::exception&) { used = 0; }
      if (used != item.size() || item.empty() || v < 0) throw std::runtime_error("--devices expects a comma-separated list of GPU ordinals, got '" + args.str("devices") + "'");
      listed.push_back(v);
    }
    if (listed.size() != numDevices) throw std::runtime_error("--devices must name one GPU per logical device (--ipus " + std::to_string(numDevices) + ")");
    deviceMap = listed;
  }
  hostGather = args.flag("host-gather");
  for (std::size_t d = 0; d < numDevices && !hostGather; ++d)
    for (std::size_t o = 0; o < d; ++o)
      if (deviceMap[o] == deviceMap[d]) {
        pt_log::info_("Logical devices {} and {} share GPU {}: HDR tiles are gathered through the host, not over RCCL", o, d, deviceMap[d]);
        hostGather = true;
        break;
      }
  const auto raysPerJob = calculateMaxRaysPerTile(imageWidth, imageHeight, geometry);
  ipuJobs.reserve(geometry.numTiles);
  for (std::size_t t = 0; t < geometry.numTiles; ++t) ipuJobs.emplace_back(raysPerJob, args, t);   // PathTracerApp.cpp:323-326
  const std::size_t jobsPerDevice = geometry.numTiles / numDevices;
  const std::size_t itemsPerDevice = raysPerJob * jobsPerDevice;
  for (std::size_t d = 0; d < numDevices; ++d) {
    pt_config cfg{};
    cfg.struct_size = sizeof(pt_config);
    // every job of the device contributes its slice and the per-tile scalars (the host half of buildGraph)
    for (std::size_t j = 0; j < jobsPerDevice; ++j)
      ipuJobs[d * jobsPerDevice + j].buildGraph(cfg, d, j * raysPerJob, geometry, args);
    cfg.max_path_length = args.u32("max-path-length");
    cfg.aa_noise_type = aaNoiseType(args.str("aa-noise-type"));
    cfg.sample_precision = PT_SAMPLES_HALF;
    cfg.device = deviceMap[d];
    // with load balancing on the resident film the devices trade image tiles: room for any deal with equal tile counts
    deviceCapacity = itemsPerDevice;
    if (args.flag("enable-load-balancing"))
      deviceCapacity = std::max(deviceCapacity, maxTileItemsPerDevice(imageWidth, imageHeight, numDevices));
    cfg.max_work_items = (std::uint32_t)deviceCapacity;
    pt_handle h = nullptr;
    if (pt_create(&cfg, &h)) throw std::runtime_error(std::string("Could not attach to device: ") + pt_last_error(nullptr));
    devices.push_back(h);
  }
  if (numDevices > 1 && hostGather) {
    pt_log::info_("HDR tiles of {} devices are gathered through the host", numDevices);
  } else if (numDevices > 1) {
    // one RCCL communicator over the devices of this process: rank d = device d, HDR tiles are gathered to rank 0
    if (pt_comm_init_all(devices.data(), (int)numDevices))
      throw std::runtime_error(std::string("Could not create the RCCL communicator: ") + pt_last_error(devices[0]));
    pt_log::info_("RCCL communicator over {} devices", numDevices);
  }
  pt_log::info_("Tracebuffer shape: [{}, {}]", ipuJobs.size(), sizeof(TraceRecord) * raysPerJob);
}

void PathTracerApp::initialiseState(std::uint32_t imageWidth, std::uint32_t imageHeight) {
  auto jobs = createTracingJobs(imageWidth, imageHeight, geometry);
  pt_log::info_("Created worklists for {} tiles", jobs.size());
  traceState = std::make_unique<PathTracerState>(imageWidth, imageHeight, jobs.size() * jobs.front().size());
  traceState->work.randomiseWorkList(jobs);
  traceState->work.getWork().active() = traceState->work.getWork().inactive();
}

// One call per device, each from its own thread (the devices work at the same time, as the IPUs of one Poplar engine do:
// PathTracerApp.cpp:205-252).  The FIRST failure is the one reported; the thread that fails asks every other device
// to abort its communicator (pt_comm_abort), so a sibling waiting inside the HDR gather for the failed rank returns at
// once with PT_ERR_COMM instead of at its deadline -- the join below can therefore never wait for ever.
template <class F>
void PathTracerApp::onEveryDevice(const char* what, F&& call) {
  std::vector<std::string> errors(devices.size());
  std::atomic<int> firstFailed{-1};
  if (devices.size() == 1) {
    if (call(0)) { errors[0] = pt_last_error(devices[0]); firstFailed = 0; }
  } else {
    std::vector<std::thread> runners;
    for (std::size_t d = 0; d < devices.size(); ++d)
      runners.emplace_back([&, d]() {
        if (!call(d)) return;
        errors[d] = pt_last_error(devices[d]);
        int none = -1;
        if (firstFailed.compare_exchange_strong(none, (int)d))
          for (std::size_t o = 0; o < devices.size(); ++o) if (o != d) pt_comm_abort(devices[o]);
      });
    for (auto& t : runners) t.join();
  }
  const int f = firstFailed.load();
  if (f >= 0) throw std::runtime_error(std::string(what) + " failed on device " + std::to_string(f) + ": " + errors[(std::size_t)f]);
}

void PathTracerApp::execute() {
  const auto imageWidth = args.u32("width"), imageHeight = args.u32("height");
  const auto seed = args.u64("seed");
  const float antiAliasingScale = args.f32("aa-noise-scale");
  const float fieldOfView = args.f32("fov") * (float)(M_PI / 180.f);      // PathTracerApp.cpp:574
  const auto steps = samplesPerPixel / samplesPerIpuStep;
  const float degrees = args.f32("env-map-rotation");
  const float radians = (degrees / 360.f) * (float)(2.0 * M_PI);          // PathTracerApp.cpp:584

  renderStartTime = std::chrono::steady_clock::now();
  pt_log::debug_("Host-phase trace ranges: {}", pt_trace::api().push ? pt_trace::api().library : "no ROCTx library found (ranges are no-ops)");

  // programs init_nif_weights and init_render_settings (PathTracerApp.cpp:612-614)
  auto initialisation = std::make_unique<pt_trace::Range>("initialisation");   // PathTracerApp.cpp:567-639
  for (std::size_t d = 0; d < devices.size(); ++d) {
    if (args.has("constant-env") && !args.str("constant-env").empty()) {
      float rgb[3] = {1, 1, 1};
      if (std::sscanf(args.str("constant-env").c_str(), "%f,%f,%f", &rgb[0], &rgb[1], &rgb[2]) != 3)
        throw std::runtime_error("--constant-env expects r,g,b");
      check(devices[d], pt_set_constant_env(devices[d], rgb), "set_constant_env");
    } else {
      models[d]->analyseModel((std::size_t)imageWidth * imageHeight / devices.size());
      models[d]->upload(devices[d]);
    }
    check(devices[d], pt_set_render_settings(devices[d], seed, antiAliasingScale, fieldOfView, radians, samplesPerIpuStep),
          "init_render_settings");
  }
  initialiseState(imageWidth, imageHeight);
  initialisation.reset();
  pt_trace::Range rendering("rendering");                                       // PathTracerApp.cpp:640

  pt_log::info_("Render started");
  // Nothing but the film ever has to leave the devices -- load balancing included: the balancer gets per-tile path-length
  // sums (kilobytes, pt_tile_costs) at the save intervals instead of the whole trace buffer every step.  The reference's
  // own loop (worklist to the host and back every step, per-item balancing) remains as --host-film, and serves the UI.
  if (args.flag("host-film") || args.u32("ui-port") != 0) executeHostFilm(steps);
  else executeResidentFilm(steps);

  auto endTime = std::chrono::steady_clock::now();
  const auto elapsedSecs = std::chrono::duration<double>(endTime - renderStartTime).count();
  pt_log::info_("Render finished: {} seconds", elapsedSecs);
  const std::size_t pixelsPerFrame = (std::size_t)imageWidth * imageHeight;
  finalSamplesPerSec = (pixelsPerFrame / elapsedSecs) * samplesPerPixel;   // PathTracerApp.cpp:786-789
  pt_log::info_("Samples/sec: {}", finalSamplesPerSec);
  pt_log::info_("Samples/sec/tile: {}", finalSamplesPerSec / ipuJobs.size());
}

void PathTracerApp::defunctState(std::uint32_t imageWidth, std::uint32_t imageHeight) {
  if (defunctTraceState) {
    defunctTraceState->film.reset();   // avoid reallocation: the worklists are large (PathTracerApp.cpp:512-515)
  } else {
    defunctTraceState.reset(new PathTracerState(imageWidth, imageHeight, traceState->work.getWork().active().size()));
  }
  // Swap and then copy the up-to-date work from the now defunct worklist:
  std::swap(traceState, defunctTraceState);
  traceState->work.getWork().active() = defunctTraceState->work.getWork().active();
  traceState->work.getWork().inactive() = defunctTraceState->work.getWork().active();
}

InterfaceServer::Status PathTracerApp::processUserInput(InterfaceServer::State& state, std::uint32_t imageWidth,
                                                        std::uint32_t imageHeight) {
  if (state.stop) {
    pt_log::info_("Rendering stopped by remote UI");
    return InterfaceServer::Status::Stop;
  }
  if (state.detach) {
    pt_log::info_("Remote UI disconnected.");   // the render just continues
    return InterfaceServer::Status::Disconnected;
  }
  if (!state.newNif.empty()) {
    pt_log::info_("Loading NIF: {}", state.newNif);
    if (loadNifModels(models.size(), state.newNif)) {
      // program init_nif_weights again (PathTracerApp.cpp:548-557): pt_upload_nif is re-callable
      for (std::size_t d = 0; d < devices.size(); ++d) models[d]->upload(devices[d]);
    }
  }
  defunctState(imageWidth, imageHeight);
  return InterfaceServer::Status::Restart;
}

void PathTracerApp::executeResidentFilm(std::uint32_t steps) {
  const auto imageWidth = args.u32("width"), imageHeight = args.u32("height");
  const float configExposure = args.f32("exposure"), configGamma = args.f32("gamma");
  const auto fileName = args.str("outfile");
  const auto saveInterval = args.u32("save-interval");
  const bool loadBalanceEnabled = args.flag("enable-load-balancing");
  pt_log::info_("Step loop: film resident on the device{}", devices.size() > 1 ? "s" : "");

  // program setup, once: every device gets its slice of the (shuffled) worklist -- pixel coordinates, zero accumulators --
  // and keeps it; with load balancing the slices are re-dealt at the save intervals (below)
  const auto& work = traceState->work.getWork().active();
  const std::size_t itemsPerDevice = work.size() / devices.size();
  const std::size_t slot = deviceCapacity;      // items per device incl. padding: equal on all devices (the gather's slot)
  const TraceRecord padding(std::numeric_limits<std::uint16_t>::max(), std::numeric_limits<std::uint16_t>::max());
  std::vector<RecordList> deviceWork(devices.size());
  for (std::size_t d = 0; d < devices.size(); ++d) {
    deviceWork[d].assign(work.begin() + d * itemsPerDevice, work.begin() + (d + 1) * itemsPerDevice);
    deviceWork[d].resize(slot, padding);
  }
  std::vector<float> tiles(devices.size() * slot * 3);   // rank 0 receives [device][slot][3] BGR film sums
  RecordList filmRecords;                                 // the gathered film as records AccumulatedImage::accumulate takes
  // Declared AFTER everything its job reads (filmRecords): if the step loop throws, unwinding destroys -- and so joins --
  // the task first, while those buffers are still alive.
  AsyncTask hostProcessing;
  onEveryDevice("setup", [&](std::size_t d) {
    pt_trace::Range r("setup");
    return pt_setup(devices[d], reinterpret_cast<const pt_trace_record*>(deviceWork[d].data()), slot);
  });
  const std::size_t nTiles = balanceTileCount(imageWidth, imageHeight);
  if (loadBalanceEnabled)
    onEveryDevice("tile costs", [&](std::size_t d) { return pt_tile_costs_enable(devices[d], kBalanceTile, kBalanceTile); });
  if (slot > itemsPerDevice)
    // padding items (u = v = 65535, LoadBalancer.cpp:66-71) are not traced (the reference's tiles trace
    // them and discard the result); the film and the tile costs skip them, Samples/sec and Rays/sec count image pixels only
    pt_log::info_("Load balancing: {} of {} work items per device are padding ({}%): room for any deal of {} image tiles", slot - itemsPerDevice,
                  slot, 100.0 * (double)(slot - itemsPerDevice) / (double)slot, nTiles);

  for (auto step = 1u; step <= steps; ++step) {
    auto loopStartTime = std::chrono::steady_clock::now();
    std::vector<pt_stats> stats(devices.size());
    // path_trace, then on the device what the host task does in the reference's loop: film += (b,g,r)/sampleCount and
    // clear the accumulators (PathTracerApp.cpp:717-745)
    onEveryDevice("Device step", [&](std::size_t d) {
      int rc;
      { pt_trace::Range r("ipu_render"); rc = pt_path_trace(devices[d]) || pt_get_stats(devices[d], &stats[d]); }   // PathTracerApp.cpp:688-699
      if (rc) return rc;
      pt_trace::Range r("accumulate_framebuffers");                                                                // :725-727, on the device
      return pt_film_accumulate(devices[d]);
    });
    std::size_t totalRays = 0;
    for (auto& s : stats) totalRays += s.segments;    // what clearInactiveAccumulators sums (LoadBalancer.cpp:198-213)
    pt_log::debug_("Path-Trace ms: {}", stats[0].path_trace_ms);
    pt_log::debug_("NIF ms: {}", stats[0].nif_ms);
    pt_log::debug_("Total ms per step: {}", stats[0].total_ms);

    if (step % saveInterval == 0 || step == steps) {
      { pt_trace::Range r("wait_for_host"); hostProcessing.waitForCompletion(); }   // the previous save still reads filmRecords (:702-706)
      // the ONE exchange of the multi-GPU path: HDR tiles to device 0 over RCCL, then to the host film
      // (--host-gather, or logical devices sharing a GPU: no communicator, every device hands its own tile to the host)
      onEveryDevice("HDR gather", [&](std::size_t d) {
        pt_trace::Range r("hdr_gather");
        if (hostGather) return pt_gather_hdr(devices[d], PT_HDR_FILM, slot, tiles.data() + d * slot * 3);
        return pt_gather_hdr(devices[d], PT_HDR_FILM, slot, d == 0 ? tiles.data() : nullptr);
      });
      filmRecords.clear();
      filmRecords.reserve(devices.size() * slot);
      for (std::size_t d = 0; d < devices.size(); ++d)
        for (std::size_t i = 0; i < slot; ++i) {
          TraceRecord r = deviceWork[d][i];
          const float* t = &tiles[(d * slot + i) * 3];
          r.b = t[0]; r.g = t[1]; r.r = t[2];
          r.sampleCount = 1;                  // accumulate() multiplies by 1 / sampleCount: the sums pass through
          filmRecords.push_back(r);
        }
      if (loadBalanceEnabled && step != steps) {
        // N3 without the worklist leaving the devices (LoadBalancer::allocateWorkByPathLength, LoadBalancer.cpp:141-192):
        // per-tile path-length sums from every device, tiles re-dealt by cost, and the film follows its pixels
        // (pt_film_seed), so every pixel's fp32 sum continues in step order: the image is bit-identical to the unbalanced one.
        pt_trace::Range balancing("run_load_balancing");                                                           // :748-751
        pt_log::info_("Load balancing started ({} image tiles over {} device{})", nTiles, devices.size(), devices.size() > 1 ? "s" : "");
        std::vector<std::vector<std::uint64_t>> part(devices.size(), std::vector<std::uint64_t>(nTiles));
        onEveryDevice("tile costs", [&](std::size_t d) { return pt_tile_costs(devices[d], part[d].data(), nTiles); });
        std::vector<std::uint64_t> cost(nTiles, 0);
        for (auto& p : part) for (std::size_t t = 0; t < nTiles; ++t) cost[t] += p[t];
        pt_log::debug_("Load balancing: {} bytes of tile costs from each device (the trace buffer is {} bytes)", nTiles * sizeof(std::uint64_t),
                       slot * sizeof(TraceRecord));
        const auto owner = dealTilesByPathLength(cost, devices.size());
        std::vector<float> running((std::size_t)imageWidth * imageHeight * 3, 0.f);   // the film so far, by pixel
        for (const auto& r : filmRecords)
          if (r.u < imageWidth && r.v < imageHeight) {
            float* p = &running[((std::size_t)r.v * imageWidth + r.u) * 3];
            p[0] = r.b; p[1] = r.g; p[2] = r.r;
          }
        std::vector<std::vector<float>> seed(devices.size());
        for (std::size_t d = 0; d < devices.size(); ++d) {
          deviceWork[d] = tileWorkList(imageWidth, imageHeight, owner, (std::int32_t)d, slot);
          seed[d].assign(slot * 3, 0.f);
          for (std::size_t i = 0; i < slot; ++i) {
            const auto& r = deviceWork[d][i];
            if (r.u < imageWidth && r.v < imageHeight)
              std::copy_n(&running[((std::size_t)r.v * imageWidth + r.u) * 3], 3, &seed[d][i * 3]);
          }
        }
        onEveryDevice("re-deal", [&](std::size_t d) {
          return pt_setup(devices[d], reinterpret_cast<const pt_trace_record*>(deviceWork[d].data()), slot) ||
                 pt_film_seed(devices[d], seed[d].data(), slot);
        });
        pt_log::info_("Load balancing finished");
      }
      hostProcessing.run([&, step]() {
        pt_trace::Range async("async_work");                                                                        // :718
        traceState->film.reset();
        traceState->film.accumulate(filmRecords);
        pt_trace::Range save("save_images");                                                                        // :761
        traceState->film.saveImages(fileName, step, configExposure, configGamma);   // hdr / step, as ever
        pt_log::info_("Saved images at step {}", step);
      });
    }

    pt_trace::Range logging("log_stats");                                                                           // :765
    auto loopEndTime = std::chrono::steady_clock::now();
    auto secs = std::chrono::duration<double>(loopEndTime - loopStartTime).count();
    const auto pixelSamplesPerStep = (double)imageWidth * imageHeight * samplesPerIpuStep;
    pt_log::info_("Completed render step {}/{} in {} seconds (Samples/sec {}) (Rays/sec {})", step, steps, secs,
                  pixelSamplesPerStep / secs, totalRays / secs);
  }
  hostProcessing.waitForCompletion();
}

void PathTracerApp::executeHostFilm(std::uint32_t steps) {
  const auto imageWidth = args.u32("width"), imageHeight = args.u32("height");
  const auto fileName = args.str("outfile");
  const bool loadBalanceEnabled = args.flag("enable-load-balancing");
  const auto saveInterval = args.u32("save-interval");
  const auto seed = args.u64("seed");
  const float antiAliasingScale = args.f32("aa-noise-scale");
  std::atomic<std::size_t> totalRays{0};   // written by the host task, read by the step log
  const std::size_t itemsPerDevice = traceState->work.getWork().active().size() / devices.size();

  // Setup remote user interface (PathTracerApp.cpp:616-633):
  std::unique_ptr<InterfaceServer> uiServer;
  InterfaceServer::State state;
  state.exposure = args.f32("exposure");
  state.gamma = args.f32("gamma");
  state.fov = args.f32("fov") * (float)(M_PI / 180.f);
  state.envRotationDegrees = args.f32("env-map-rotation");
  state.interactiveSamples = args.u32("interactive-samples");
  if (const auto uiPort = args.u32("ui-port")) {
    uiServer.reset(new InterfaceServer((int)uiPort));
    uiServer->setInitialState(state);
    uiServer->start();
    uiServer->initialiseVideoStream(imageWidth, imageHeight);
  }
  // Declared AFTER the server, the state and the counter its job uses: if anything in the loop throws, unwinding joins
  // the task (its destructor) before the InterfaceServer it talks to is destroyed.
  AsyncTask hostProcessing;
  pt_log::info_("Step loop: the reference's (worklist to the host and back every step, host film)");
  constexpr std::size_t sampleCountReversionStep = 5;
  std::uint32_t lastStep = 0;   // steps accumulated into the current film
  auto sendRenderSettings = [&]() {   // program init_render_settings (PathTracerApp.cpp:678-686)
    const float radians = (state.envRotationDegrees / 360.f) * (float)(2.0 * M_PI);
    for (auto h : devices)
      check(h, pt_set_render_settings(h, seed, antiAliasingScale, state.fov, radians, samplesPerIpuStep), "init_render_settings");
  };

  for (auto step = 1u; step <= steps; ++step) {
    auto loopStartTime = std::chrono::steady_clock::now();

    // Do the simple thing and restart the entire render if any state changed (PathTracerApp.cpp:656-676):
    if (uiServer && uiServer->stateChanged()) {
      pt_trace::Range r("ui_processing");                                     // PathTracerApp.cpp:653
      state = uiServer->consumeState();
      const auto status = processUserInput(state, imageWidth, imageHeight);
      // (the previous step's host task may still be sending to the client: join it before the server goes away -- the
      // reference resets the server while that task can still be running)
      if (status == InterfaceServer::Status::Stop) {
        hostProcessing.waitForCompletion();
        uiServer.reset();
        break;
      }
      if (status == InterfaceServer::Status::Disconnected) {
        hostProcessing.waitForCompletion();
        uiServer.reset();
      } else if (status == InterfaceServer::Status::Restart) {
        renderStartTime = loopStartTime;   // the final Samples/sec counts from the restart (:669)
        step = 1;
        samplesPerIpuStep = state.interactiveSamples;
      }
    } else {
      // (no `uiServer &&` here, as in the reference: a client that restarts the render and then detaches or closes must
      // not leave the device on the interactive sample count for the rest of the run)
      if (step == sampleCountReversionStep) {
        // No UI input for a few steps so revert to performant number of samples:
        samplesPerIpuStep = args.u32("samples-per-step");
        pt_log::debug_("Interaction stopped reverting samples per step to: {}", samplesPerIpuStep);
      }
    }
    // Render settings can only be updated on these steps (:678-686), server or no server:
    if (step == 1 || step == sampleCountReversionStep) { pt_trace::Range r("update_ipu_settings"); sendRenderSettings(); }

    // setup -> path_trace -> read_results on every device (PathTracerApp.cpp:692-694).  The worklist is
    // cut into equal contiguous slices, one per device, as tiles are cut over IPUs.
    auto& active = traceState->work.getWork().active();
    std::vector<pt_stats> stats(devices.size());
    onEveryDevice("Device step", [&](std::size_t d) {
      auto* slice = reinterpret_cast<pt_trace_record*>(active.data() + d * itemsPerDevice);
      pt_handle h = devices[d];
      pt_trace::Range r("ipu_render");                                        // :688-699
      return pt_setup(h, slice, itemsPerDevice) || pt_path_trace(h) || pt_read_results(h, slice, itemsPerDevice, &stats[d]);
    });
    pt_log::debug_("Path-Trace ms: {}", stats[0].path_trace_ms);
    pt_log::debug_("NIF ms: {}", stats[0].nif_ms);
    pt_log::debug_("Total ms per step: {}", stats[0].total_ms);
    pt_log::debug_("Step {} took {} samples per pixel from sample index {}", step, samplesPerIpuStep, stats[0].first_sample);   // (pt_stats.paths counts real pixels only: not padding)

    const auto deviceDone = std::chrono::steady_clock::now();
    { pt_trace::Range r("wait_for_host"); hostProcessing.waitForCompletion(); }   // join the previous async task before swapping (:703-708)
    traceState->work.getWork().swap();
    pt_log::debug_("Device calls (setup + path_trace + read_results) wall ms: {}",
                   std::chrono::duration<double, std::milli>(deviceDone - loopStartTime).count());
    pt_log::debug_("Waited for host film task ms: {}",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - deviceDone).count());

    // The work list and film are captured by pointer: user interaction may make them defunct while this runs (:717)
    hostProcessing.run([&, step, workPtr = &traceState->work, filmPtr = &traceState->film]() {
      pt_trace::Range async("async_work");                                    // :718
      { pt_trace::Range r("accumulate_framebuffers"); filmPtr->accumulate(workPtr->getWork().inactive()); }   // :725-727
      if (uiServer) {
        const auto ui = uiServer->getState();
        {
          pt_trace::Range r("tone_map");                                      // :732-734 (+ ui_encode_video: the preview is sent as text here)
          uiServer->sendPreviewImage(filmPtr->updateLdrImage(step, ui.exposure, ui.gamma));
        }
        pt_trace::Range r("ui_send_events");                                  // :738
        uiServer->updateProgress((int)step, (int)steps);
      }
      if (loadBalanceEnabled && step > 1) { pt_trace::Range r("run_load_balancing"); workPtr->allocateWorkByPathLength(ipuJobs); }   // :748-751
      { pt_trace::Range r("clear_accumulators"); totalRays = workPtr->clearInactiveAccumulators(); }                                   // :753-755
      if (step % saveInterval == 0 || step == steps) {
        if (uiServer) {
          // with a UI attached the raw image is transmitted at the save interval instead of saved (:748-753)
          uiServer->startSendingRawImage(filmPtr->getHdrImage(), step);
        } else {
          pt_trace::Range r("save_images");                                   // :761
          filmPtr->saveImages(fileName, step, state.exposure, state.gamma);
          pt_log::info_("Saved images at step {}", step);
        }
      }
    });

    auto loopEndTime = std::chrono::steady_clock::now();
    auto secs = std::chrono::duration<double>(loopEndTime - loopStartTime).count();
    const auto pixelSamplesPerStep = (double)imageWidth * imageHeight * samplesPerIpuStep;
    pt_log::info_("Completed render step {}/{} in {} seconds (Samples/sec {}) (Rays/sec {})", step, steps, secs,
                  pixelSamplesPerStep / secs, totalRays.load() / secs);
    if (uiServer) uiServer->updateSampleRate((float)(pixelSamplesPerStep / secs), (float)(totalRays.load() / secs));
    lastStep = step;
  }
  hostProcessing.waitForCompletion();
  if (args.u32("ui-port") != 0 && lastStep) {
    // the film the client saw last is also left on disk, so an interactive session ends with an image like a batch run
    traceState->film.saveImages(fileName, lastStep, state.exposure, state.gamma);
    pt_log::info_("Saved images at step {}", lastStep);
  }
}
