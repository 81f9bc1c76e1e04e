// ptmi.hip -- implementation of include/ptmi.h.  One translation unit: this file holds the entry points (create / upload /
// setup / path_trace / read_results ...); ptmi_context.h the per-handle state, ptmi_nif_pack.h the NIF normalisation and weight
// packing, ptmi_nif_launch.h the kernel launchers, ptmi_film_comm.h the resident film, tile costs and the RCCL hand-off.
//
// Build (see __graft_entry__.build): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// -ffp-contract=off is part of the numerical contract (pt_device_math.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <limits.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ptmi.h"
#include "pt_nif.h"
#include "pt_nif_gemm.h"
#include "pt_nif_f32.h"
#include "pt_trace.h"

#include "ptmi_comm_worker.h"
#include "ptmi_context.h"
#include "ptmi_nif_pack.h"
#include "ptmi_nif_launch.h"
#ifdef PTMI_DIAG_BUILD
#include "diag/ptmi_trace_variants.h"
#endif

static void comm_release(pt_handle h);
static void forget_replay_state(pt_handle h);   // ptmi_film_comm.h: orderly end of the handle's communicator (finalize, polled; abort on expiry)

extern "C" {

int pt_abi_version(void) { return PTMI_ABI_VERSION; }

const char* pt_last_error(pt_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int pt_create(const pt_config* cfg, pt_handle* out) {
  if (!cfg || !out) { g_create_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  if (cfg->struct_size != sizeof(pt_config)) { g_create_error = "pt_config.struct_size mismatch"; return PT_ERR_INVALID_ARGUMENT; }
  if (cfg->width == 0 || cfg->height == 0 || cfg->width > 65535 || cfg->height > 65535) {
    g_create_error = "width/height must be in 1..65535 (TraceRecord coordinates are uint16)";
    return PT_ERR_INVALID_ARGUMENT;
  }
  if (cfg->max_path_length == 0 || cfg->max_path_length > 64) { g_create_error = "max_path_length must be in 1..64"; return PT_ERR_INVALID_ARGUMENT; }
  if (cfg->roulette_depth == 0) { g_create_error = "roulette_depth must be >= 1 (0 is undefined behaviour in the reference)"; return PT_ERR_INVALID_ARGUMENT; }
  if (!(cfg->stop_prob >= 0.f && cfg->stop_prob < 1.f)) { g_create_error = "stop_prob must be in [0,1)"; return PT_ERR_INVALID_ARGUMENT; }
  if (cfg->aa_noise_type < 0 || cfg->aa_noise_type > 2) { g_create_error = "invalid aa_noise_type"; return PT_ERR_INVALID_ARGUMENT; }
  if (cfg->max_work_items == 0) { g_create_error = "max_work_items must be > 0"; return PT_ERR_INVALID_ARGUMENT; }

  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    g_create_error = "no HIP device available: the MI355X path has no CPU fallback";
    return PT_ERR_NO_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= count) { g_create_error = "device ordinal out of range"; return PT_ERR_INVALID_ARGUMENT; }

  pt_handle h = new pt_context();
  h->cfg = *cfg;
  auto bail = [&](int code) { g_create_error = h->error; pt_destroy(h); return code; };
#define PT_HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->error = std::string(#call) + ": " + hipGetErrorString(e_); return bail(hip_status(e_)); } } while (0)
  PT_HIPC(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  PT_HIPC(hipGetDeviceProperties(&prop, cfg->device));
  h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // The NIF stream outranks the trace stream: when NIF(b) and trace(b+1) become ready together, the NIF kernel's 256
  // CU-sized workgroups must be placed first and the small trace workgroups fill what is left, not the reverse.
  int prio_least = 0, prio_greatest = 0;
  PT_HIPC(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; }
  else { PT_HIPC(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_greatest)); h->own_stream = true; }

  const uint32_t n = cfg->max_work_items;
  h->capacity = n;
  uint32_t k = cfg->iterations_per_batch;
  // auto: as many sample iterations per batch as memory allows, 32 at most.  Both sets of batch buffers together cost
  // kBatchBytesPerPath per path of capacity; the budget is 24 GiB or a quarter of what is free on the device, whichever is
  // smaller -- small against 288 GB of HBM3E.  Larger batches mean fewer launch tails (10 instead of 43 NIF launches per
  // 300-spp step of the 1104 x 1000 image bought 2 %) and fewer passes over the accumulators (a 3840 x 2160 image at the
  // "32 M paths" of rounds 1-4 took 4 iterations per batch: 250 accumulate launches per 1000-spp step, each re-reading and
  // re-writing 32 B of accumulators per item for 52 B of payload).
  if (k == 0) {
    size_t free_b = 0, total_b = 0;
    uint64_t budget = 24ull << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 < budget) budget = free_b / 4;
    const uint64_t paths = budget / pt_context::kBatchBytesPerPath;
    k = (uint32_t)std::min<uint64_t>(32, std::max<uint64_t>(1, paths / n));
  }
  if ((uint64_t)k * n >= (1ull << 31)) k = (uint32_t)(((1ull << 31) - 1) / n);
  if (k == 0) { h->error = "max_work_items too large"; return bail(PT_ERR_INVALID_ARGUMENT); }
  h->iters_per_batch = k;
#ifdef PTMI_DIAG_BUILD
  if (const char* e = getenv("PTMI_FIRST_BATCH")) h->first_batch_iters = (uint32_t)std::max(1, atoi(e));   // tuning sweep of the profiling build; the product uses 1
#endif
  h->batch_paths_cap = (size_t)k * n;

  PT_HIPC(dev_alloc(&h->d_records, n));
  PT_HIPC(dev_alloc(&h->acc.pix, n));
  PT_HIPC(dev_alloc(&h->acc.r, n));
  PT_HIPC(dev_alloc(&h->acc.g, n));
  PT_HIPC(dev_alloc(&h->acc.b, n));
  PT_HIPC(dev_alloc(&h->acc.count, n));
  PT_HIPC(dev_alloc(&h->acc.length, n));
  PT_HIPC(dev_alloc(&h->d_counters, 3));   // segments, escaped (per step); real work items (per pt_setup)
  PT_HIPC(hipHostMalloc(reinterpret_cast<void**>(&h->h_counters), 3 * sizeof(unsigned long long), hipHostMallocDefault));
  h->trace_blocks = std::min<uint32_t>((uint32_t)ptd::kMaxRegions, (uint32_t)pt_context::kTraceBlocksPerCu * (uint32_t)h->n_cus);
#ifdef PTMI_DIAG_BUILD
  if (const char* e = getenv("PTMI_TRACE_BLOCKS")) h->trace_blocks = std::min<uint32_t>((uint32_t)ptd::kMaxRegions, (uint32_t)std::max(1, atoi(e)));   // grid-size sweep of the profiling build
#endif
  const TraceGrid g = trace_grid((uint32_t)h->batch_paths_cap, h->trace_blocks);
  h->queue_cap = (size_t)g.blocks * g.region_cap;
  for (auto& B : h->bb) {
    PT_HIPC(dev_alloc(&B.q_u, h->queue_cap));
    PT_HIPC(dev_alloc(&B.q_v, h->queue_cap));
    PT_HIPC(dev_alloc(&B.q_tr, h->queue_cap));
    PT_HIPC(dev_alloc(&B.q_tg, h->queue_cap));
    PT_HIPC(dev_alloc(&B.q_tb, h->queue_cap));
    PT_HIPC(dev_alloc(&B.q_path, h->queue_cap));
    PT_HIPC(dev_alloc(&B.survivors, h->queue_cap));
    PT_HIPC(dev_alloc(&B.states, 3 * h->queue_cap));
    PT_HIPC(dev_alloc(&B.region_count, (size_t)ptd::kMaxRegions));
    PT_HIPC(dev_alloc(&B.plen, h->batch_paths_cap));
    PT_HIPC(dev_alloc(&B.rad_r, h->batch_paths_cap));
    PT_HIPC(dev_alloc(&B.rad_g, h->batch_paths_cap));
    PT_HIPC(dev_alloc(&B.rad_b, h->batch_paths_cap));
    PT_HIPC(hipEventCreateWithFlags(&B.traced, hipEventDisableTiming));
    PT_HIPC(hipEventCreateWithFlags(&B.accumulated, hipEventDisableTiming));
  }
  PT_HIPC(hipEventCreateWithFlags(&h->chunk_fork, hipEventDisableTiming));
  for (int i = 0; i + 1 < pt_context::kChunkSets; ++i) {
    PT_HIPC(hipStreamCreateWithPriority(&h->chunk_stream[i], hipStreamNonBlocking, prio_greatest));
    PT_HIPC(hipEventCreateWithFlags(&h->chunk_join[i], hipEventDisableTiming));
  }
  PT_HIPC(hipStreamCreateWithPriority(&h->trace_stream, hipStreamNonBlocking, prio_least));
  PT_HIPC(hipStreamCreateWithPriority(&h->acc_stream, hipStreamNonBlocking, prio_least));
#ifdef PTMI_DIAG_BUILD
  // profiling build: PTMI_SERIAL=1 runs the trace and accumulate kernels on the NIF stream (no overlap at all): every
  // stage's HIP-event time is then that kernel alone on the device
  if (getenv("PTMI_SERIAL")) {
    (void)hipStreamDestroy(h->trace_stream); h->trace_stream = h->stream;
    (void)hipStreamDestroy(h->acc_stream); h->acc_stream = h->stream;
    h->serial = true;
  }
#endif
#undef PT_HIPC
  *out = h;
  return PT_OK;
}

int pt_destroy(pt_handle h) {
  if (!h) return PT_OK;
  (void)hipSetDevice(h->cfg.device);   // ipu_trace --ipus N holds handles of several devices in one process
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  // the communicator goes first, while the streams it has worked on still exist: a non-blocking communicator's teardown is
  // asynchronous too (ncclCommFinalize, polled against the handle's deadline; ncclCommAbort if it does not finish)
  comm_release(h);
  free_batch_buffers(h);
  (void)hipFree(h->d_records);
  (void)hipFree(h->acc.pix); (void)hipFree(h->acc.r); (void)hipFree(h->acc.g); (void)hipFree(h->acc.b); (void)hipFree(h->acc.count); (void)hipFree(h->acc.length);
  (void)hipFree(h->d_counters);
  if (h->h_counters) (void)hipHostFree(h->h_counters);
  (void)hipFree(h->d_wpack); (void)hipFree(h->d_bpack);
  (void)hipFree(h->d_gemm_act[0]); (void)hipFree(h->d_gemm_act[1]); (void)hipFree(h->d_gemm_feat); (void)hipFree(h->d_tile_start);
  (void)hipFree(h->d_stamps);
  (void)hipFree(h->d_head_partial); (void)hipFree(h->d_head_in);
  (void)hipFree(h->d_f32_weights); (void)hipFree(h->d_f32_act[0]); (void)hipFree(h->d_f32_act[1]); (void)hipFree(h->d_f32_feat);
  (void)hipFree(h->d_scratch);
  (void)hipFree(h->d_hdr_stage); (void)hipFree(h->d_hdr_gather); (void)hipFree(h->d_film);
  (void)hipFree(h->d_slot_check);
  (void)hipFree(h->tiles.cost); (void)hipFree(h->d_tile_tmp);
  for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
  if (h->trace_stream && !h->serial) { (void)hipStreamSynchronize(h->trace_stream); (void)hipStreamDestroy(h->trace_stream); }
  if (h->acc_stream && !h->serial) { (void)hipStreamSynchronize(h->acc_stream); (void)hipStreamDestroy(h->acc_stream); }
  for (int i = 0; i + 1 < pt_context::kChunkSets; ++i) {
    if (h->chunk_stream[i]) { (void)hipStreamSynchronize(h->chunk_stream[i]); (void)hipStreamDestroy(h->chunk_stream[i]); }
    if (h->chunk_join[i]) (void)hipEventDestroy(h->chunk_join[i]);
  }
  if (h->chunk_fork) (void)hipEventDestroy(h->chunk_fork);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PT_OK;
}

// A model with float32 layers: the float path (pt_nif_f32.h).  All layers float32: every matmul, bias add and ReLU in float, as
// the reference gives a matmul its kernel's type (NifModel.cpp:314-325).  A model that MIXES float32 and binary16 layers is an
// EXTENSION with cast points of this library's own (the reference ships no such model and never casts x between layers): a
// binary16 layer is widened exactly (fp16 -> fp32 is lossless, and a product of two halves is exact in float), its sum is
// rounded to half and its bias added in half, and the activations it reads are cast to half first -- the fp32 FMA chain in k
// order is then the same sum the fp16 kernels' oracle forms.  Such a model runs at the fp32 matrix rate throughout.
static int upload_nif_f32(pt_handle h, const pt_layer* layers, uint32_t n_layers, uint32_t embedding_dim, float max,
                          const float mean[3], int32_t log_tonemap) {
  std::vector<HostLayerF32> L(n_layers);
  uint64_t flops = 0;
  for (uint32_t l = 0; l < n_layers; ++l) {
    L[l].rows = layers[l].rows; L[l].cols = layers[l].cols; L[l].relu = layers[l].relu != 0; L[l].has_bias = layers[l].bias != nullptr;
    L[l].f16 = layers[l].dtype == PT_DTYPE_F16;
    const size_t count = (size_t)L[l].rows * L[l].cols;
    if (L[l].f16) {
      const uint16_t* kp = static_cast<const uint16_t*>(layers[l].kernel);
      L[l].kernel.resize(count);
      for (size_t i = 0; i < count; ++i) L[l].kernel[i] = host_h2f(kp[i]);
      if (layers[l].bias) {
        const uint16_t* bp = static_cast<const uint16_t*>(layers[l].bias);
        L[l].bias.resize(L[l].cols);
        for (uint32_t i = 0; i < L[l].cols; ++i) L[l].bias[i] = host_h2f(bp[i]);
      }
    } else {
      const float* kp = static_cast<const float*>(layers[l].kernel);
      L[l].kernel.assign(kp, kp + count);
      if (layers[l].bias) { const float* bp = static_cast<const float*>(layers[l].bias); L[l].bias.assign(bp, bp + L[l].cols); }
    }
    flops += 2ull * L[l].rows * L[l].cols + (layers[l].bias ? L[l].cols : 0);  // NifModel.cpp:129-133
  }
  std::vector<float> blob;
  std::vector<pt_context::F32Layer> F;
  uint32_t Hp = 0, Ep = 0;
  if (int rc = pack_nif_f32(h, L, embedding_dim, blob, F, Hp, Ep)) return rc;
  PT_HIP(hipSetDevice(h->cfg.device));
  PT_HIP(hipStreamSynchronize(h->stream));
  h->nif_valid = false;
  const uint32_t chunk = 4096;   // queue tiles per chunk (131,072 samples; a multiple of 8: a workgroup takes eight): 2 x 168 MB of activations at width 320
  for (float** p : {&h->d_f32_weights, &h->d_f32_act[0], &h->d_f32_act[1], &h->d_f32_feat}) {
    if (*p) PT_HIP(hipFree(*p));
    *p = nullptr;
  }
  h->f32_chunk = 0;
  PT_HIP(dev_alloc(&h->d_f32_weights, blob.size()));
  PT_HIP(hipMemcpy(h->d_f32_weights, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
  h->f32_act_set = (size_t)chunk * 32 * Hp;                 // one buffer set per chunk in flight: pt_context::chunk_stream
  h->f32_feat_set = (size_t)chunk * 32 * 4 * Ep;
  PT_HIP(dev_alloc(&h->d_f32_act[0], pt_context::kChunkSets * h->f32_act_set));
  PT_HIP(dev_alloc(&h->d_f32_act[1], pt_context::kChunkSets * h->f32_act_set));
  PT_HIP(dev_alloc(&h->d_f32_feat, pt_context::kChunkSets * h->f32_feat_set));
  if (!h->d_tile_start) PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_tile_start), (ptd::kMaxRegions + 1) * 4));
  const HostLayerF32& head = L[n_layers - 1];
  for (int o = 0; o < 3; ++o) h->head_bias[o] = head.has_bias ? head.bias[o] : 0.f;
  ptd::NifParams N;
  memset(&N, 0, sizeof(N));
  N.n_layers = n_layers;
  N.n_freq = embedding_dim;
  N.max = max;
  N.mean0 = mean[0]; N.mean1 = mean[1]; N.mean2 = mean[2];
  N.log_tonemap = log_tonemap;
  h->nif = N;
  h->f32_layers = F;
  h->f32_chunk = chunk;
  h->f32_lda = Hp;
  h->f32_ldf = 4 * Ep;
  h->nif_hidden = (int)Hp;
  h->nif_emb = (int)Ep;
  h->nif_f32 = true;
  h->nif_gemm = false;
  h->nif_flops = flops;
  h->nif_valid = true;
  h->env_const = false;
  forget_replay_state(h);
  return PT_OK;
}

int pt_upload_nif(pt_handle h, const pt_layer* layers, uint32_t n_layers, uint32_t embedding_dim, float max,
                  const float mean[3], int32_t log_tonemap) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!layers || !mean || n_layers == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "null NIF arguments");
  std::vector<HostLayer> L(n_layers);
  uint64_t flops = 0;
  bool any_f32 = false;
  for (uint32_t l = 0; l < n_layers; ++l) {
    if (layers[l].dtype != PT_DTYPE_F16 && layers[l].dtype != PT_DTYPE_F32)
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF weights must be float16 or float32");
    if (!layers[l].kernel || layers[l].rows == 0 || layers[l].cols == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "empty layer kernel");
    any_f32 = any_f32 || layers[l].dtype == PT_DTYPE_F32;
  }
  if (any_f32) return upload_nif_f32(h, layers, n_layers, embedding_dim, max, mean, log_tonemap);   // each layer in its own type
  for (uint32_t l = 0; l < n_layers; ++l) {
    L[l].rows = layers[l].rows;
    L[l].cols = layers[l].cols;
    L[l].relu = layers[l].relu != 0;
    const size_t count = (size_t)L[l].rows * L[l].cols;
    {
      const uint16_t* kp = static_cast<const uint16_t*>(layers[l].kernel);
      L[l].kernel.assign(kp, kp + count);
      if (layers[l].bias) {
        const uint16_t* bp = static_cast<const uint16_t*>(layers[l].bias);
        L[l].bias.assign(bp, bp + L[l].cols);
      }
    }
    flops += 2ull * L[l].rows * L[l].cols + (layers[l].bias ? L[l].cols : 0);  // NifModel.cpp:129-133
  }
  std::vector<HostLayer> padded;
  NifPlan plan;
  int rc = normalize_nif(h, L, embedding_dim, padded, plan);
  if (rc) return rc;
  std::vector<uint16_t> wpack, bpack;
  ptd::NifParams N;
  std::vector<float> head_in;
  float head_bias[3] = {0, 0, 0};
  uint32_t head_piece_base = 0;
  rc = plan.gemm ? pack_nif_g16(h, padded, plan.Ep, wpack, bpack, N, head_in, head_bias, head_piece_base)
                 : pack_nif(h, padded, plan.Ep, wpack, bpack, N);
  if (rc) return rc;
  PT_HIP(hipSetDevice(h->cfg.device));
  PT_HIP(hipStreamSynchronize(h->stream));
  if (h->d_wpack) PT_HIP(hipFree(h->d_wpack));
  if (h->d_bpack) PT_HIP(hipFree(h->d_bpack));
  h->d_wpack = nullptr; h->d_bpack = nullptr;
  h->nif_valid = false;
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_wpack), wpack.size() * 2));
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_bpack), bpack.size() * 2));
  PT_HIP(hipMemcpy(h->d_wpack, wpack.data(), wpack.size() * 2, hipMemcpyHostToDevice));
  PT_HIP(hipMemcpy(h->d_bpack, bpack.data(), bpack.size() * 2, hipMemcpyHostToDevice));
  N.wpack = h->d_wpack;
  N.bpack = h->d_bpack;
  N.n_freq = plan.E;
  N.max = max;
  N.mean0 = mean[0]; N.mean1 = mean[1]; N.mean2 = mean[2];
  N.log_tonemap = log_tonemap;
  if (plan.gemm) {   // wide network: chunk buffers of the layer-by-layer path
    uint32_t chunk = 4096;
#ifdef PTMI_DIAG_BUILD
    if (const char* c = getenv("PTMI_GEMM_CHUNK")) chunk = (uint32_t)atoi(c) / 8u * 8u;   // chunk-size sweep of the profiling build
    if (chunk == 0) chunk = 8;
#endif
    // activations: Hp / 32 k-steps x 2 KiB per queue tile; features: ceil(Ep / 8) k-steps x 2 KiB (the 32-shape kernels of the
    // profiling build use Ep / 4 pieces of 1 KiB and may read one piece past an odd count: the larger of the two + slack)
    const size_t act_bytes = (size_t)chunk * (plan.Hp / 16) * 1024, feat_bytes = (size_t)chunk * ((plan.Ep + 7) / 8) * 2048 + 2048;
    for (int i = 0; i < 2; ++i) {
      if (h->d_gemm_act[i]) PT_HIP(hipFree(h->d_gemm_act[i]));
      h->d_gemm_act[i] = nullptr;
      h->gemm_chunk = 0;
      PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_gemm_act[i]), pt_context::kChunkSets * act_bytes));   // one buffer set per chunk in flight: pt_context::chunk_stream
    }
    h->gemm_act_set = act_bytes / 16;
    if (h->d_gemm_feat) PT_HIP(hipFree(h->d_gemm_feat));
    h->d_gemm_feat = nullptr;
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_gemm_feat), pt_context::kChunkSets * feat_bytes));
    h->gemm_feat_set = feat_bytes / 16;
    if (!h->d_tile_start) PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_tile_start), (ptd::kMaxRegions + 1) * 4));
    // fused head: partial sums of the 2 x (Hp / 256) slices, and the head's feature-input weights if it concatenates them
    if (h->d_head_partial) PT_HIP(hipFree(h->d_head_partial));
    h->d_head_partial = nullptr;
    h->head_partial_set = (size_t)2 * (plan.Hp / 256) * chunk * 32;
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_head_partial), pt_context::kChunkSets * h->head_partial_set * sizeof(float4)));
    if (h->d_head_in) PT_HIP(hipFree(h->d_head_in));
    h->d_head_in = nullptr;
    if (!head_in.empty()) {
      PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_head_in), head_in.size() * sizeof(float)));
      PT_HIP(hipMemcpy(h->d_head_in, head_in.data(), head_in.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    memcpy(h->head_bias, head_bias, sizeof(head_bias));
    h->head_piece_base = head_piece_base;
#ifdef PTMI_DIAG_BUILD
    if (!h->d_stamps) { PT_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_stamps), 256 * 8)); PT_HIP(hipMemset(h->d_stamps, 0, 256 * 8)); }
#endif
    h->gemm_chunk = chunk;
  }
  h->nif = N;
  h->nif_hidden = (int)plan.Hp;
  h->nif_emb = (int)plan.Ep;
  h->nif_f32 = false;
  h->nif_gemm = plan.gemm;
  h->nif_flops = flops;
  h->nif_valid = true;
  h->env_const = false;
  forget_replay_state(h);
  return PT_OK;
}

int pt_set_constant_env(pt_handle h, const float rgb[3]) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!rgb) return fail(h, PT_ERR_INVALID_ARGUMENT, "null rgb");
  h->env_const = true;
  memcpy(h->env_rgb, rgb, 12);
  forget_replay_state(h);
  return PT_OK;
}

int pt_set_render_settings(pt_handle h, uint64_t seed, float aa, float fov, float azimuth, uint32_t spp) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (spp == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "samples_per_step must be > 0");
  // TraceRecord::sampleCount is uint16 (TraceRecord.hpp:10) and the host divides by it (AccumulatedImage.cpp:69-71):
  // more than 65535 samples per step would wrap it (to 0 at 65536).  pathLength (:11) is uint16 too and is allowed
  // to wrap as in the reference (samples_per_step x max_path_length may exceed 65535); stats.segments is exact.
  if (spp > 65535u) return fail(h, PT_ERR_INVALID_ARGUMENT, "samples_per_step must be <= 65535 (TraceRecord::sampleCount is uint16)");
  if (!(fov > 0.f && fov < 3.14159f)) return fail(h, PT_ERR_INVALID_ARGUMENT, "fov must be in (0, pi) radians");
  if (!h->settings_valid || seed != h->seed) h->sample_cursor = 0;
  h->seed = seed; h->aa_scale = aa; h->fov = fov; h->azimuth = azimuth; h->samples_per_step = spp;
  h->settings_valid = true;
  return PT_OK;
}

int pt_setup(pt_handle h, const pt_trace_record* work, size_t n) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!work && n) return fail(h, PT_ERR_INVALID_ARGUMENT, "null worklist");
  if (n > h->capacity) return fail(h, PT_ERR_INVALID_ARGUMENT, "worklist larger than max_work_items");
  PT_HIP(hipSetDevice(h->cfg.device));
  h->n_items = (uint32_t)n;
  h->film_steps = 0;
  if (h->d_film) PT_HIP(hipMemsetAsync(h->d_film, 0, (size_t)h->capacity * 12, h->stream));   // a new worklist starts a new film
  if (h->tiles.n_tiles) PT_HIP(hipMemsetAsync(h->tiles.cost, 0, (size_t)h->tiles.n_tiles * 8, h->stream));   // ... and new per-tile sums
  h->n_real = 0;
  if (n == 0) return PT_OK;
  static_assert(sizeof(pt_trace_record) == 20 && sizeof(ptd::TraceRecordDev) == 20, "TraceRecord wire format");
  PT_HIP(hipMemcpyAsync(h->d_records, work, n * sizeof(pt_trace_record), hipMemcpyHostToDevice, h->stream));
  PT_HIP(hipMemsetAsync(h->d_counters + 2, 0, sizeof(unsigned long long), h->stream));
  hipLaunchKernelGGL(ptd::unpack_records_kernel, dim3(((uint32_t)n + 255) / 256), dim3(256), 0, h->stream, h->d_records,
                     (uint32_t)n, h->acc, h->cfg.width, h->cfg.height, h->d_counters + 2);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpyAsync(h->h_counters + 2, h->d_counters + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  PT_HIP(hipStreamSynchronize(h->stream));  // host buffer is not touched after return
  h->n_real = (uint32_t)h->h_counters[2];   // items that are not padding: the only ones that are traced
  return PT_OK;
}

// AccumulateContributions for one batch: four consecutive items per thread where the worklist's size allows it.
static void launch_accumulate(pt_handle h, uint32_t n, uint32_t iters, const pt_context::BatchBuffers& B) {
  if (n % 4u == 0)
    hipLaunchKernelGGL(ptd::accumulate4_kernel, dim3((n / 4u + 255) / 256), dim3(256), 0, h->acc_stream, n, iters, B.plen, B.rad_r,
                       B.rad_g, B.rad_b, h->acc, h->d_counters);
  else
    hipLaunchKernelGGL(ptd::accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, h->acc_stream, n, iters, B.plen, B.rad_r,
                       B.rad_g, B.rad_b, h->acc, h->d_counters);
}

// Enqueue the whole path_trace program on the three streams.  Returns at the first failing call; the caller drains the
// streams either way, so a failure in the middle of the batch loop never leaves kernels running on buffers the host
// is about to reuse or free.
using StageSpan = pt_context::StageSpan;

static void forget_replay_state(pt_handle h) {   // pt_calibrate_nif may only replay a batch of the most recent NIF step
  for (auto& B : h->bb) { B.last_paths = 0; B.last_regions = 0; B.last_region_cap = 0; }
}

static int enqueue_path_trace(pt_handle h, std::vector<StageSpan>& spans, size_t& e_begin_i, size_t& e_end_i) {
  const uint32_t n = h->n_items;
  forget_replay_state(h);
  PT_HIP(hipMemsetAsync(h->d_counters, 0, 2 * sizeof(unsigned long long), h->stream));
  ptd::TraceParams P;
  fill_trace_params(h, P);
  P.n_items = n;
  // Schedule: T(b) on trace_stream, N(b) on stream, A(b) on acc_stream.  T(b+1) overlaps N(b) (VALU under MFMA),
  // A(b) overlaps N(b+1) (a short HBM-bound pass), so the NIF launches follow each other without a gap;
  // buffer set b&1 is reused by T(b+2) once A(b) has consumed it.  The A(b) are ordered among themselves (one
  // stream), which keeps every pixel's fp32 sum in iteration order.
  size_t ev = 0;
  e_begin_i = ev;
  hipEvent_t e_begin = get_event(h, ev++);
  if (!e_begin) return fail(h, PT_ERR_HIP, "hipEventCreate failed");
  PT_HIP(hipEventRecord(e_begin, h->stream));
  PT_HIP(hipStreamWaitEvent(h->trace_stream, e_begin, 0));   // counters memset and earlier work on `stream`
  // Batches.  The first batch is kept short when the step has several: its trace kernel is the only one with no NIF kernel
  // to hide under, so the sooner it ends the sooner the MFMA pipes start.  The remaining iterations are dealt EVENLY over as
  // few batches as the capacity allows (sizes differ by one at most), so no step ends on a stub of a batch whose launch
  // tails weigh as much as a full one's.  Per-pixel sums stay in iteration order whatever the split.
  const uint32_t spp = h->samples_per_step;
  const uint32_t first = (!h->env_const && spp > 2u * h->iters_per_batch) ? std::min(h->first_batch_iters, h->iters_per_batch) : 0u;
  const uint32_t rest = spp - first;
  const uint32_t rest_batches = (rest + h->iters_per_batch - 1u) / h->iters_per_batch;
  const uint32_t base = rest_batches ? rest / rest_batches : 0u, longer = rest_batches ? rest % rest_batches : 0u;
  uint32_t done = 0, batch = 0;
  while (done < spp) {
    const uint32_t r = batch - (first ? 1u : 0u);                       // index among the evenly dealt batches
    const uint32_t iters = (first && batch == 0) ? first : base + (r < longer ? 1u : 0u);
    const uint32_t total = iters * n;
    const TraceGrid g = trace_grid(total, h->trace_blocks);
    pt_context::BatchBuffers& B = h->bb[batch & 1];
    P.sample_base = h->sample_cursor + done;
    P.total_paths = total;
    P.n_waves = g.n_waves;
    P.region_cap = g.region_cap;
    bind_batch(P, B);

    hipEvent_t t0 = get_event(h, ev), t1 = get_event(h, ev + 1), n0 = get_event(h, ev + 2), n1 = get_event(h, ev + 3),
               a1 = get_event(h, ev + 4), a0 = get_event(h, ev + 5);
    if (!t0 || !t1 || !n0 || !n1 || !a1 || !a0) return fail(h, PT_ERR_HIP, "hipEventCreate failed");
    if (batch >= 2) PT_HIP(hipStreamWaitEvent(h->trace_stream, B.accumulated, 0));
    PT_HIP(hipEventRecord(t0, h->trace_stream));
#ifdef PTMI_DIAG_BUILD
    if (!launch_trace_variant(h, P, g))   // A/B switches of the profiling build (diag/ptmi_trace_variants.h), read per launch
#endif
    hipLaunchKernelGGL(ptd::trace_kernel, dim3(g.blocks), dim3(ptd::kTraceBlock), 0, h->trace_stream, P);
    PT_HIP(hipGetLastError());
    PT_HIP(hipEventRecord(t1, h->trace_stream));
    PT_HIP(hipEventRecord(B.traced, h->trace_stream));
    spans.push_back({ev, ev + 1, 0});
    PT_HIP(hipStreamWaitEvent(h->stream, B.traced, 0));
    PT_HIP(hipEventRecord(n0, h->stream));
    if (!h->env_const) {
      ptd::NifParams N = h->nif;
      N.q_u = B.q_u; N.q_v = B.q_v; N.q_tr = B.q_tr; N.q_tg = B.q_tg; N.q_tb = B.q_tb; N.q_path = B.q_path;
      N.region_count = B.region_count;
      N.n_regions = g.blocks;
      N.region_cap = g.region_cap;
      N.rad_r = B.rad_r; N.rad_g = B.rad_g; N.rad_b = B.rad_b;
      N.out_bgr = nullptr;
#ifdef PTMI_DIAG_BUILD
      // Fault injection for the error-path test (tests/test_gpu_edge_cases.py), test build only: pt_diag_inject_fault(h, b)
      // makes batch b's NIF launch report a failure after the earlier batches are already queued on all three streams.
      if (h->diag_fault_batch >= 0 && (uint32_t)h->diag_fault_batch == batch)
        return fail(h, PT_ERR_HIP, "injected fault: NIF launch of batch " + std::to_string(batch));
#endif
      if (int rc = launch_nif(h, N, h->n_cus)) return rc;
      B.last_paths = total; B.last_regions = g.blocks; B.last_region_cap = g.region_cap;
      spans.push_back({ev + 2, ev + 3, 1});
      h->stats.nif_launches += 1;
    }
    PT_HIP(hipEventRecord(n1, h->stream));
    PT_HIP(hipStreamWaitEvent(h->acc_stream, n1, 0));
    PT_HIP(hipEventRecord(a0, h->acc_stream));
    launch_accumulate(h, n, iters, B);
    PT_HIP(hipGetLastError());
    PT_HIP(hipEventRecord(a1, h->acc_stream));
    PT_HIP(hipEventRecord(B.accumulated, h->acc_stream));
    spans.push_back({ev + 5, ev + 4, 2});
    ev += 6;
    h->stats.trace_launches += 1;
    h->stats.accumulate_launches += 1;
    done += iters;
    batch += 1;
  }
  hipEvent_t e_acc = get_event(h, ev++);
  if (!e_acc) return fail(h, PT_ERR_HIP, "hipEventCreate failed");
  PT_HIP(hipEventRecord(e_acc, h->acc_stream));
  PT_HIP(hipStreamWaitEvent(h->stream, e_acc, 0));          // later work on `stream` sees the accumulated film
  // the two counters travel to pinned host memory on the stream itself: the host waits once, for e_end
  PT_HIP(hipMemcpyAsync(h->h_counters, h->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  e_end_i = ev;
  hipEvent_t e_end = get_event(h, ev++);
  if (!e_end) return fail(h, PT_ERR_HIP, "hipEventCreate failed");
  PT_HIP(hipEventRecord(e_end, h->stream));
  return PT_OK;
}

// Per-stage device times of the last step from its event pairs; read on demand (pt_get_stats, pt_read_results,
// pt_calibrate_nif -- which reuses two of the events).  The events are complete: pt_path_trace returned after e_end.
static int resolve_stage_times(pt_handle h) {
  if (!h->spans_pending) return PT_OK;
  h->spans_pending = false;
  PT_HIP(hipSetDevice(h->cfg.device));
  for (const StageSpan& s : h->spans) {
    float ms = 0.f;
    PT_HIP(hipEventElapsedTime(&ms, h->events[s.a], h->events[s.b]));
    if (s.kind == 0) h->stats.path_trace_ms += ms;
    else if (s.kind == 1) h->stats.nif_ms += ms;
    else h->stats.accumulate_ms += ms;
  }
  float total_ms = 0.f;
  PT_HIP(hipEventElapsedTime(&total_ms, h->events[h->e_begin_i], h->events[h->e_end_i]));
  h->stats.total_ms = total_ms;
  return PT_OK;
}

int pt_path_trace(pt_handle h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!h->settings_valid) return fail(h, PT_ERR_NOT_READY, "pt_set_render_settings has not been called");
  if (!h->env_const && !h->nif_valid) return fail(h, PT_ERR_NOT_READY, "no environment: call pt_upload_nif or pt_set_constant_env");
  PT_HIP(hipSetDevice(h->cfg.device));
  memset(&h->stats, 0, sizeof(h->stats));
  h->spans_pending = false;
  h->stats.nif_flops_per_sample = h->env_const ? 0 : h->nif_flops;
  h->stats.first_sample = h->sample_cursor;
  const uint32_t n = h->n_items;
  if (n == 0) return PT_OK;

  h->spans.clear();
  const int rc = enqueue_path_trace(h, h->spans, h->e_begin_i, h->e_end_i);
  // Success: everything the step queued -- the trace kernels (through the NIF stream's waits), the chunk streams (joined),
  // the accumulate passes (e_acc), the counters' copy -- is ordered before the end of `stream`: ONE wait.  (Rounds 1-4
  // synchronised five streams, copied the counters with a blocking hipMemcpy and read 3 event pairs per batch here: host
  // time that a 1 M-path step of BASELINE configs[0] paid in full.)  Failure: drain every stream, whatever was queued.
  hipError_t s0 = hipStreamSynchronize(h->stream), s1 = hipSuccess, s2 = hipSuccess, s3 = hipSuccess;
  if (rc || s0 != hipSuccess) {
    for (hipStream_t cs : h->chunk_stream) { const hipError_t e = hipStreamSynchronize(cs); if (e != hipSuccess) s3 = e; }
    s1 = hipStreamSynchronize(h->trace_stream);
    s2 = hipStreamSynchronize(h->acc_stream);
    (void)hipStreamSynchronize(h->stream);   // (what joined it late is drained too)
  }
  if (rc) return rc;   // h->error names the failing call
  for (hipError_t e : {s0, s1, s2, s3})
    if (e != hipSuccess) return fail(h, PT_ERR_HIP, std::string("path_trace: ") + hipGetErrorString(e));
  h->sample_cursor += h->samples_per_step;
  h->stats.paths = (uint64_t)h->n_real * h->samples_per_step;   // padding items are not traced (pt_setup)
  h->stats.segments = h->h_counters[0];
  h->stats.escaped = h->h_counters[1];
  h->spans_pending = true;
  return PT_OK;
}

int pt_nif_kernel_name(pt_handle h, char* buf, size_t n) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!buf || n == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "null name buffer");
  snprintf(buf, n, "%s", h->nif_kernel.c_str());
  return PT_OK;
}

// The NIF stage of the last path_trace's largest batch again, alone on the device: same kernel, same queue (it is still in
// the batch buffers), same output arrays (rewritten with the same values; the accumulators are not touched).
int pt_calibrate_nif(pt_handle h, uint32_t launches, double* ms_per_launch, uint64_t* evaluations) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!ms_per_launch || !evaluations || launches == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "pt_calibrate_nif: bad arguments");
  if (!h->nif_valid) return fail(h, PT_ERR_NOT_READY, "pt_upload_nif has not been called");
  const pt_context::BatchBuffers& B = h->bb[h->bb[1].last_paths > h->bb[0].last_paths ? 1 : 0];
  if (B.last_paths == 0) return fail(h, PT_ERR_NOT_READY, "no path_trace with a NIF environment has run on this handle since the last pt_upload_nif / pt_set_constant_env");
  PT_HIP(hipSetDevice(h->cfg.device));
  if (int rc = resolve_stage_times(h)) return rc;   // two of the step's events are reused below
  PT_HIP(hipStreamSynchronize(h->stream));
  std::vector<uint32_t> counts(B.last_regions);
  PT_HIP(hipMemcpy(counts.data(), B.region_count, counts.size() * 4, hipMemcpyDeviceToHost));
  uint64_t evals = 0;
  for (uint32_t c : counts) evals += c;
  ptd::NifParams N = h->nif;
  N.q_u = B.q_u; N.q_v = B.q_v; N.q_tr = B.q_tr; N.q_tg = B.q_tg; N.q_tb = B.q_tb; N.q_path = B.q_path;
  N.region_count = B.region_count;
  N.n_regions = B.last_regions;
  N.region_cap = B.last_region_cap;
  N.rad_r = B.rad_r; N.rad_g = B.rad_g; N.rad_b = B.rad_b;
  N.out_bgr = nullptr;
  hipEvent_t e0 = get_event(h, 0), e1 = get_event(h, 1);
  if (!e0 || !e1) return fail(h, PT_ERR_HIP, "hipEventCreate failed");
  if (int rc = launch_nif(h, N, h->n_cus)) return rc;                      // untimed: clocks and caches as inside a step
  PT_HIP(hipEventRecord(e0, h->stream));
  int rc = PT_OK;
  for (uint32_t i = 0; i < launches && rc == PT_OK; ++i) rc = launch_nif(h, N, h->n_cus);
  PT_HIP(hipEventRecord(e1, h->stream));
  hipError_t s3 = hipSuccess;
  for (hipStream_t cs : h->chunk_stream) { const hipError_t e = hipStreamSynchronize(cs); if (e != hipSuccess) s3 = e; }
  PT_HIP(hipStreamSynchronize(h->stream));
  if (rc) return rc;
  if (s3 != hipSuccess) return fail(h, PT_ERR_HIP, std::string("pt_calibrate_nif: ") + hipGetErrorString(s3));
  float ms = 0.f;
  PT_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = (double)ms / launches;
  *evaluations = evals;
  return PT_OK;
}

int pt_get_stats(pt_handle h, pt_stats* stats) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!stats) return fail(h, PT_ERR_INVALID_ARGUMENT, "null stats");
  if (int rc = resolve_stage_times(h)) return rc;
  *stats = h->stats;
  return PT_OK;
}

int pt_read_results(pt_handle h, pt_trace_record* work, size_t n, pt_stats* stats) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!work && n) return fail(h, PT_ERR_INVALID_ARGUMENT, "null worklist");
  if (n != h->n_items) return fail(h, PT_ERR_INVALID_ARGUMENT, "worklist size differs from the one given to pt_setup");
  PT_HIP(hipSetDevice(h->cfg.device));
  if (n) {
    hipLaunchKernelGGL(ptd::pack_records_kernel, dim3(((uint32_t)n + 255) / 256), dim3(256), 0, h->stream, h->d_records,
                       (uint32_t)n, h->acc);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(work, h->d_records, n * sizeof(pt_trace_record), hipMemcpyDeviceToHost, h->stream));
    PT_HIP(hipStreamSynchronize(h->stream));
  }
  if (stats) {
    if (int rc = resolve_stage_times(h)) return rc;
    *stats = h->stats;
  }
  return PT_OK;
}

int pt_export_hdr_device(pt_handle h, void* device_bgr, size_t n) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!device_bgr || n != h->n_items) return fail(h, PT_ERR_INVALID_ARGUMENT, "bad export buffer");
  PT_HIP(hipSetDevice(h->cfg.device));
  if (n) hipLaunchKernelGGL(ptd::export_hdr_kernel, dim3(((uint32_t)n + 255) / 256), dim3(256), 0, h->stream, (uint32_t)n, h->acc,
                            static_cast<float*>(device_bgr));
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int pt_clear_accumulators(pt_handle h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  PT_HIP(hipSetDevice(h->cfg.device));
  if (h->n_items) hipLaunchKernelGGL(ptd::clear_accum_kernel, dim3((h->n_items + 255) / 256), dim3(256), 0, h->stream, h->n_items, h->acc);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int pt_synchronize(pt_handle h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  PT_HIP(hipSetDevice(h->cfg.device));
  PT_HIP(hipStreamSynchronize(h->stream));
  return PT_OK;
}

int pt_nif_infer(pt_handle h, const float* u, const float* v, size_t n, float* bgr) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!h->nif_valid) return fail(h, PT_ERR_NOT_READY, "pt_upload_nif has not been called");
  if (n == 0) return PT_OK;
  if (!u || !v || !bgr) return fail(h, PT_ERR_INVALID_ARGUMENT, "null buffer");
  if (n >= (1ull << 31)) return fail(h, PT_ERR_INVALID_ARGUMENT, "too many samples");
  PT_HIP(hipSetDevice(h->cfg.device));
  const size_t bytes = n * 4 * 2 + n * 12 + 16;
  int rc = ensure_scratch(h, bytes);
  if (rc) return rc;
  float* d_u = static_cast<float*>(h->d_scratch);
  float* d_v = d_u + n;
  float* d_out = d_v + n;
  uint32_t* d_count = reinterpret_cast<uint32_t*>(d_out + 3 * n);
  const uint32_t cnt = (uint32_t)n;
  PT_HIP(hipMemcpyAsync(d_u, u, n * 4, hipMemcpyHostToDevice, h->stream));
  PT_HIP(hipMemcpyAsync(d_v, v, n * 4, hipMemcpyHostToDevice, h->stream));
  PT_HIP(hipMemcpyAsync(d_count, &cnt, 4, hipMemcpyHostToDevice, h->stream));
  ptd::NifParams N = h->nif;
  N.q_u = d_u; N.q_v = d_v;
  N.q_tr = N.q_tg = N.q_tb = nullptr; N.q_path = nullptr;
  N.region_count = d_count;
  N.n_regions = 1;
  N.region_cap = cnt;
  N.rad_r = N.rad_g = N.rad_b = nullptr;
  N.out_bgr = d_out;
  rc = launch_nif(h, N, h->n_cus);
  if (rc) return rc;
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpyAsync(bgr, d_out, n * 12, hipMemcpyDeviceToHost, h->stream));
  PT_HIP(hipStreamSynchronize(h->stream));
  return PT_OK;
}

int pt_trace_paths(pt_handle h, const uint16_t* u, const uint16_t* v, const uint32_t* sample_index, size_t n,
                   pt_path_record* out) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!h->settings_valid) return fail(h, PT_ERR_NOT_READY, "pt_set_render_settings has not been called");
  if (n == 0) return PT_OK;
  if (!u || !v || !sample_index || !out) return fail(h, PT_ERR_INVALID_ARGUMENT, "null buffer");
  PT_HIP(hipSetDevice(h->cfg.device));
  static_assert(sizeof(pt_path_record) == sizeof(ptd::PathRecordOut), "pt_path_record layout");
  const size_t bytes = n * (2 + 2 + 4) + 16 + n * sizeof(pt_path_record);
  int rc = ensure_scratch(h, bytes);
  if (rc) return rc;
  char* base = static_cast<char*>(h->d_scratch);
  ptd::PathRecordOut* d_out = reinterpret_cast<ptd::PathRecordOut*>(base);
  uint32_t* d_s = reinterpret_cast<uint32_t*>(base + n * sizeof(pt_path_record));
  uint16_t* d_u = reinterpret_cast<uint16_t*>(d_s + n);
  uint16_t* d_v = d_u + n;
  PT_HIP(hipMemcpyAsync(d_s, sample_index, n * 4, hipMemcpyHostToDevice, h->stream));
  PT_HIP(hipMemcpyAsync(d_u, u, n * 2, hipMemcpyHostToDevice, h->stream));
  PT_HIP(hipMemcpyAsync(d_v, v, n * 2, hipMemcpyHostToDevice, h->stream));
  ptd::TraceParams P;
  fill_trace_params(h, P);
  hipLaunchKernelGGL(ptd::trace_paths_kernel, dim3(((uint32_t)n + 127) / 128), dim3(128), 0, h->stream, P, d_u, d_v, d_s,
                     (uint32_t)n, d_out);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpyAsync(out, d_out, n * sizeof(pt_path_record), hipMemcpyDeviceToHost, h->stream));
  PT_HIP(hipStreamSynchronize(h->stream));
  return PT_OK;
}

}  // extern "C"

#include "ptmi_film_comm.h"

extern "C" {

#ifdef PTMI_DIAG_BUILD
// test build only: the NIF launch of batch `batch` of every following pt_path_trace fails (batch < 0: off)
int pt_diag_inject_fault(pt_handle h, int32_t batch) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  h->diag_fault_batch = batch;
  return PT_OK;
}

// test build only: the point-to-point half of pt_gather_hdr on a communicator of ONE rank -- a grouped ncclSend / ncclRecv of
// `floats` floats from this rank to itself on the non-blocking communicator, with the same polled waits as the gather.  A
// one-GPU box cannot run two ranks (RCCL refuses two ranks on one device); this is as much of the send / receive path as it
// can execute.  out_ok = 1 if the received buffer equals the sent one.
int pt_diag_comm_self_exchange(pt_handle h, size_t floats, int* out_ok) {
  if (!h || !out_ok || floats == 0) return PT_ERR_INVALID_ARGUMENT;
  if (!h->comm || h->comm_world != 1) return fail(h, PT_ERR_INVALID_ARGUMENT, "needs a communicator of one rank");
  PT_HIP(hipSetDevice(h->cfg.device));
  float *src = nullptr, *dst = nullptr;
  PT_HIP(dev_alloc(&src, floats));
  PT_HIP(dev_alloc(&dst, floats));
  std::vector<float> host(floats);
  for (size_t i = 0; i < floats; ++i) host[i] = (float)(i % 977) * 0.25f - 3.f;
  PT_HIP(hipMemcpy(src, host.data(), floats * 4, hipMemcpyHostToDevice));
  PT_HIP(hipMemset(dst, 0, floats * 4));
  const auto deadline = comm_deadline(h);
  ncclResult_t r = ncclGroupStart();
  if (r == ncclSuccess || r == ncclInProgress) r = ncclSend(src, floats, ncclFloat, 0, h->comm, h->stream);
  if (r == ncclSuccess || r == ncclInProgress) r = ncclRecv(dst, floats, ncclFloat, 0, h->comm, h->stream);
  const ncclResult_t e = ncclGroupEnd();
  if (r != ncclSuccess && r != ncclInProgress) return comm_fail(h, std::string("self exchange: ") + ncclGetErrorString(r));
  if (e != ncclSuccess && e != ncclInProgress) return comm_fail(h, std::string("self exchange: ncclGroupEnd: ") + ncclGetErrorString(e));
  if (int rc = comm_wait_host(h, "self exchange", deadline)) return rc;
  if (int rc = comm_wait_stream(h, "self exchange", deadline)) return rc;
  std::vector<float> back(floats);
  PT_HIP(hipMemcpy(back.data(), dst, floats * 4, hipMemcpyDeviceToHost));
  *out_ok = memcmp(back.data(), host.data(), floats * 4) == 0;
  (void)hipFree(src); (void)hipFree(dst);
  return PT_OK;
}

// profiling build only: in-kernel clock of the last stamped fused-NIF launch (nif_kernel_v3 with DIAG bit 5):
// out2[0] = shader cycles, out2[1] = 100 MHz ticks of its workgroup 0
int pt_diag_nif_clock(pt_handle h, unsigned long long* out2) {
  if (!h || !out2) return PT_ERR_INVALID_ARGUMENT;
  PT_HIP(hipSetDevice(h->cfg.device));
  PT_HIP(hipStreamSynchronize(h->stream));
  PT_HIP(hipMemcpyFromSymbol(out2, HIP_SYMBOL(ptd::g_nif_clock), 16));
  return PT_OK;
}

// profiling build only: copy the 256 phase stamps of the last stamped layer launch (PTMI_GEMM_DIAG=64)
int pt_diag_stamps(pt_handle h, unsigned long long* out256) {
  if (!h || !h->d_stamps || !out256) return PT_ERR_INVALID_ARGUMENT;
  PT_HIP(hipSetDevice(h->cfg.device));
  PT_HIP(hipStreamSynchronize(h->stream));
  PT_HIP(hipMemcpy(out256, h->d_stamps, 256 * 8, hipMemcpyDeviceToHost));
  return PT_OK;
}
#endif

int pt_comm_info(pt_handle h, int* rank, int* world) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (rank) *rank = h->comm_rank;
  if (world) *world = h->comm_world;
  return PT_OK;
}

}  // extern "C"
