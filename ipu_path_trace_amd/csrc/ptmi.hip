// ptmi.hip -- implementation of include/ptmi.h: device state, weight packing, kernel launches.
//
// Build (see __graft_entry__.build): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// -ffp-contract=off is part of the numerical contract (pt_device_math.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ptmi.h"
#include "pt_nif.h"
#include "pt_nif_gemm.h"
#include "pt_nif_f32.h"
#ifdef PTMI_DIAG_BUILD
#include "diag/pt_nif_gemm32.h"
#include "diag/pt_nif16.h"
#include "diag/pt_nif_variants.h"
#endif
#include "pt_trace.h"
#ifdef PTMI_DIAG_BUILD
#include "diag/pt_trace_v1.h"
#endif

namespace {

thread_local std::string g_create_error;

// ---- binary16 helpers on the host (weights arrive as raw fp16 bytes)
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

uint16_t host_f2h(float f) {
  uint32_t x = f2u(f), sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (ax < 0x33000001u) return (uint16_t)sign;
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u, shift, hexp;
  if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); hexp = 0; } else { shift = 13; hexp = (uint32_t)(e + 15); }
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) q += 1u;
  uint32_t h = hexp == 0 ? q : ((hexp - 1u) << 10) + q;
  return (uint16_t)(sign | h);
}

float host_h2f(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    float v = (float)m * 5.9604644775390625e-08f;
    return sign ? -v : v;
  }
  if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

inline float host_hround(float f) { return host_h2f(host_f2h(f)); }

struct HostLayer {
  uint32_t rows, cols;
  std::vector<uint16_t> kernel;  // [rows][cols]
  std::vector<uint16_t> bias;    // [cols] or empty
  bool relu;
};

}  // namespace

struct pt_context {
  pt_config cfg{};
  std::string error;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cus = 256;

  // worklist
  uint32_t n_items = 0;
  uint32_t capacity = 0;
  ptd::TraceRecordDev* d_records = nullptr;
  ptd::Accum acc{};
  unsigned long long* d_counters = nullptr;  // [0] segments, [1] escaped

  // batch buffers, double-buffered: the trace kernel of batch b+1 runs on `trace_stream` while the NIF
  // kernel of batch b (MFMA-bound) runs on `stream`
  uint32_t iters_per_batch = 1;
  uint32_t first_batch_iters = 1;   // iterations of a step's first batch (see enqueue_path_trace)
  size_t batch_paths_cap = 0;
  size_t queue_cap = 0;
  struct BatchBuffers {
    float *q_u = nullptr, *q_v = nullptr, *q_tr = nullptr, *q_tg = nullptr, *q_tb = nullptr;
    uint32_t* q_path = nullptr;
    uint4* survivors = nullptr;        // primary-phase notes of the trace kernel, one region per trace workgroup
    float4* states = nullptr;          // path states after the first shading, three planes of queue_cap float4
    uint32_t* region_count = nullptr;
    uint8_t* plen = nullptr;
    float *rad_r = nullptr, *rad_g = nullptr, *rad_b = nullptr;
    hipEvent_t traced = nullptr;       // trace kernel of the batch using this set has finished
    hipEvent_t accumulated = nullptr;  // accumulate kernel has consumed this set
  } bb[2];
  hipStream_t trace_stream = nullptr;
  hipStream_t acc_stream = nullptr;   // accumulate(b) runs here, so NIF(b+1) follows NIF(b) back to back on `stream`
  bool serial = false;   // profiling build only: trace kernels share the NIF stream

  // render settings
  bool settings_valid = false;
  uint64_t seed = 0;
  float aa_scale = 0, fov = 0, azimuth = 0;
  uint32_t samples_per_step = 0;
  uint32_t sample_cursor = 0;  // absolute index of the next sample iteration

  // environment
  bool env_const = false;
  float env_rgb[3] = {0, 0, 0};
  bool nif_valid = false;
  int nif_hidden = 0, nif_emb = 0;   // PADDED hidden width / embedding dimension the kernels are instantiated for
  bool nif_gemm = false;  // layer-by-layer path (pt_nif_gemm.h)
  bool nif_gemm32 = false;   // profiling build: the round-2 32x32x16 layer kernels (diag/pt_nif_gemm32.h) for the A/B
  // float32 models (pt_nif_f32.h): padded row-major kernels and biases of all layers in one buffer, chunk buffers
  bool nif_f32 = false;
  struct F32Layer { size_t w_off, b_off; uint32_t k_act, k_in, ldw, relu; };
  std::vector<F32Layer> f32_layers;
  float* d_f32_weights = nullptr;
  float* d_f32_act[2] = {nullptr, nullptr};
  float* d_f32_feat = nullptr;
  uint32_t f32_chunk = 0, f32_lda = 0, f32_ldf = 0;
  float4* d_head_partial = nullptr;   // fused head: [2 FB][chunk samples] partial sums
  float4* d_head_in = nullptr;        // head weights of the Fourier-feature inputs [4][E], if the head concatenates them
  float head_bias[3] = {0, 0, 0};
  uint32_t head_piece_base = 0;
  bool nif_m16 = false;   // weights packed for nif_kernel_v4 (16x16x32 MFMA) rather than the 32x32x16 kernels
  ptd::NifParams nif{};
  uint4* d_wpack = nullptr;
  uint4* d_bpack = nullptr;
  uint64_t nif_flops = 0;
  // layer-by-layer path of the wide networks (pt_nif_gemm.h): activation ping-pong and feature pieces of one chunk
  uint4* d_gemm_act[2] = {nullptr, nullptr};
  uint4* d_gemm_feat = nullptr;
  uint32_t* d_tile_start = nullptr;
  uint32_t gemm_chunk = 0;   // 32-sample tiles per chunk (multiple of 8); 0 = path not set up
  // The layer-by-layer paths run the chunks of a queue round-robin on the NIF stream and on extra ones (chunk_stream):
  // chunks are independent, so one chunk's layer launch fills the CUs another's is draining (the ramp / drain / gap of a
  // launch is ~3-4 % of a 240 us layer).  Every chunk buffer therefore exists kChunkSets times (set s at offset s x *_set).
  static constexpr int kChunkSets = 2;                  // chunks in flight (C5: 1 -> 2 streams +2.7 % on one box, 0 on another; 3: -1 %)
  hipStream_t chunk_stream[kChunkSets - 1] = {};        // sets 1.. (set 0 runs on the NIF stream itself)
  hipEvent_t chunk_fork = nullptr, chunk_join[kChunkSets - 1] = {};
  int chunk_sets = kChunkSets;                           // profiling build: PTMI_CHUNK_STREAMS lowers it for the A/B
  size_t gemm_act_set = 0, gemm_feat_set = 0, head_partial_set = 0;   // uint4 / uint4 / float4 elements per set
  size_t f32_act_set = 0, f32_feat_set = 0;                            // floats per set
  unsigned long long* d_stamps = nullptr;   // profiling build: 256 phase stamps of the wide-NIF layer kernel
  int diag_fault_batch = -1;                // test build: batch whose NIF launch fails (pt_diag_inject_fault), -1 = none

  // stats
  pt_stats stats{};
  std::vector<hipEvent_t> events;

  // scratch for the standalone entry points
  void* d_scratch = nullptr;
  size_t scratch_bytes = 0;

  // multi-GPU film hand-off: RCCL communicator (one rank per handle) and the HDR tile buffers
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  bool comm_broken = false;                  // the communicator was aborted (deadline, peer failure, pt_comm_abort): gathers fail until a new one is made
  std::atomic<bool> comm_abort_req{false};   // pt_comm_abort from another thread: the polling loops see it and abort
  uint32_t comm_timeout_ms = 120000;         // deadline of every communicator operation (pt_comm_set_timeout)
  size_t comm_slot_agreed = 0;               // slot_items value every rank of the communicator is known to use
  long long* d_slot_check = nullptr;         // {slot, -slot} for the agreement all-reduce
  float* d_film = nullptr;         // resident film: [capacity][3] BGR, sum over steps of the per-step means
  ptd::TileGrid tiles{};           // per-tile path-length sums for the balancer (pt_tile_costs_enable), n_tiles = 0: off
  unsigned long long* d_tile_tmp = nullptr;   // tracked sums + current accumulators, staged for the copy to the host
  uint32_t film_steps = 0;
  float* d_hdr_stage = nullptr;    // this rank's tile: [slot_items][3] mean BGR, zero padded
  size_t hdr_stage_floats = 0;
  float* d_hdr_gather = nullptr;   // root only: [world][slot_items][3]
  size_t hdr_gather_floats = 0;
};

namespace {

#define PT_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      h->error = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return PT_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

int fail(pt_handle h, int code, const std::string& msg) {
  h->error = msg;
  return code;
}

template <typename T>
hipError_t dev_alloc(T** p, size_t count) {
  return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
}

int ensure_scratch(pt_handle h, size_t bytes) {
  if (bytes <= h->scratch_bytes) return PT_OK;
  if (h->d_scratch) PT_HIP(hipFree(h->d_scratch));
  h->d_scratch = nullptr;
  h->scratch_bytes = 0;
  PT_HIP(hipMalloc(&h->d_scratch, bytes));
  h->scratch_bytes = bytes;
  return PT_OK;
}

// Scene constants of src/codelets/codelets.cpp:111-144.
void fill_scene(ptd::TraceParams& P) {
  const float gain = 2.f;  // :127
  struct Src { int disc; float c[3]; float r; float col[3]; int type; };
  const Src src[ptd::kNumObjects] = {
      {0, {-1.8575f, -0.98714f, -3.6f}, 0.6f, {1.f * gain, .89f * gain, .55f * gain}, ptd::MAT_DIFFUSE},          // :112,:128,:137
      {0, {0.74795f, -0.55f, -4.3816f}, 1.05f, {1.f, 1.f, 1.f}, ptd::MAT_SPECULAR},                                // :113,:138
      {0, {1.9929f, -1.08666f, (float)-3.23}, 0.5f, {0.75f, 0.75f, 0.75f}, ptd::MAT_REFRACTIVE},                   // :114,:131,:139
      {0, {(float)-0.19931, -1.183f, -2.75f}, 0.4f, {.8f * gain, .06f * gain, .391f * gain}, ptd::MAT_DIFFUSE},   // :115,:129,:140
      {0, {(float)-0.19931, -1.183f, -2.75f}, 0.4001f, {1.f, 1.f, 1.f}, ptd::MAT_REFRACTIVE},                      // :116,:141
      {1, {0.f, -1.6f, -5.22f}, 3.5f, {.98f * gain, .76f * gain, .66f * gain}, ptd::MAT_DIFFUSE},                  // :121,:130,:143
  };
  for (int i = 0; i < ptd::kNumObjects; ++i) {
    ptd::SceneObject& o = P.obj[i];
    o.cx = src[i].c[0]; o.cy = src[i].c[1]; o.cz = src[i].c[2];
    o.radius = src[i].r;
    o.r2 = src[i].r * src[i].r;
    o.nx = 0.f; o.ny = src[i].disc ? 1.f : 0.f; o.nz = 0.f;
    o.colr = src[i].col[0]; o.colg = src[i].col[1]; o.colb = src[i].col[2];
    o.type = src[i].type;
    o.is_disc = src[i].disc;
  }
}

void fill_trace_params(pt_handle h, ptd::TraceParams& P) {
  memset(&P, 0, sizeof(P));
  fill_scene(P);
  const pt_config& c = h->cfg;
  const float w = (float)c.width, hgt = (float)c.height;
  const float fov = host_hround(h->fov);        // field_of_view stream is half (PathTracerApp.cpp:591)
  P.width_f = w;
  P.height_f = hgt;
  P.tx = tanf(fov * 0.5f);                      // light::pixelToRay (INFERRED: DESIGN.md, camera model)
  P.ty = (hgt / w) * P.tx;
  P.aa_scale = host_hround(h->aa_scale);        // anti_alias_scale stream is half (:590)
  P.stop_prob = host_hround(c.stop_prob);       // IpuPathTraceJob.cpp:137
  P.rr_factor = 1.0f / (1.0f - P.stop_prob);
  P.ri = host_hround(c.refractive_index);       // IpuPathTraceJob.cpp:133
  P.azimuth = h->azimuth;
  P.seed_lo = (uint32_t)h->seed;
  P.seed_hi = (uint32_t)(h->seed >> 32);
  P.max_path_length = c.max_path_length;
  P.roulette_depth = c.roulette_depth;
  P.aa_type = c.aa_noise_type;
  P.samples_half = (c.sample_precision == PT_SAMPLES_HALF);
  P.env_const = h->env_const ? 1 : 0;
  P.env_r = h->env_rgb[0]; P.env_g = h->env_rgb[1]; P.env_b = h->env_rgb[2];
  P.pix = h->acc.pix;
  P.state_stride = h->queue_cap;
}

void bind_batch(ptd::TraceParams& P, const pt_context::BatchBuffers& B) {
  P.q_u = B.q_u; P.q_v = B.q_v; P.q_tr = B.q_tr; P.q_tg = B.q_tg; P.q_tb = B.q_tb; P.q_path = B.q_path;
  P.region_count = B.region_count;
  P.survivors = B.survivors;
  P.states = B.states;
  P.plen = B.plen;
  P.rad_r = B.rad_r; P.rad_g = B.rad_g; P.rad_b = B.rad_b;
}

// Trace-grid geometry for a batch of `total` paths.
struct TraceGrid {
  uint32_t blocks, n_waves, region_cap;
};
TraceGrid trace_grid(uint32_t total) {
  const uint32_t n_chunks = (total + 63u) / 64u;
  uint32_t blocks = (n_chunks + 3u) / 4u;
  if (blocks > (uint32_t)ptd::kMaxRegions) blocks = ptd::kMaxRegions;
  if (blocks == 0) blocks = 1;
  TraceGrid g;
  g.blocks = blocks;
  g.n_waves = blocks * 4u;
  g.region_cap = 4u * ((n_chunks + g.n_waves - 1u) / g.n_waves) * 64u;
  return g;
}

// ---- NIF shape normalisation ---------------------------------------------------------------
// The reference builds whatever Dense stack the H5 describes (NifModel.cpp:295-326).  The MFMA kernels want a uniform
// hidden width (a multiple of 32 for the register-resident kernels, of 256 for the layer-by-layer path) and 4 | E, so
// the stack is zero-padded to that: a padded output feature has zero weights and zero bias (its activation is 0 with
// or without ReLU), a padded input row multiplies it by zero, and a padded frequency slot (E not a multiple of 4) has
// zero weights and a zero feature (NifParams::n_freq).  Arithmetic on the true entries is unchanged.
struct NifPlan {
  uint32_t E = 0, Ep = 0;   // frequencies per coordinate: true / padded to a multiple of 4
  uint32_t Hp = 0;          // padded uniform hidden width
  bool gemm = false;        // layer-by-layer path (pt_nif_gemm.h) instead of the register-resident kernels
};

constexpr uint32_t kMaxFusedHidden = 320;    // nif_kernel_v3/v2: two activation vectors of H halves per sample in VGPRs
constexpr uint32_t kMaxGemmHidden = 2048;    // nifg_layer_kernel: bias tiles of one layer in 4 KiB of LDS

int normalize_nif(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<HostLayer>& out, NifPlan& plan) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E > 16) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be in 1..16");
  const uint32_t in_dim = 4 * E, Ep = (E + 3u) / 4u * 4u, in_p = 4 * Ep;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  uint32_t widest = 0;
  for (uint32_t l = 0; l + 1 < n; ++l) widest = std::max(widest, L[l].cols);
  plan.E = E;
  plan.Ep = Ep;
  plan.gemm = widest > kMaxFusedHidden;
  plan.Hp = plan.gemm ? (widest + 255u) / 256u * 256u : (widest + 31u) / 32u * 32u;
  if (plan.Hp > kMaxGemmHidden) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden layers wider than 2048 are not supported");
  const uint32_t Hp = plan.Hp;
  out.assign(n, HostLayer());
  uint32_t prev = 0;   // true width of the previous layer's output
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    bool concat = false;
    if (l == 0) {
      // (rows == in_dim checked above)
    } else if (Y.rows == prev) {
    } else if (Y.rows == prev + in_dim) {   // NifModel.cpp:305-308: x = concat(x, input) when the widths differ
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer " + std::to_string(l) + ": input width " + std::to_string(Y.rows) +
                                                   " is neither the previous layer's width nor that plus the 4*embedding features");
    }
    HostLayer& Z = out[l];
    const bool feats = (l == 0) || concat;
    const uint32_t act_p = l ? Hp : 0u, act_t = l ? prev : 0u;
    Z.rows = act_p + (feats ? in_p : 0u);
    Z.cols = (l + 1 == n) ? 3u : Hp;
    Z.relu = Y.relu;
    Z.kernel.assign((size_t)Z.rows * Z.cols, 0);
    for (uint32_t r = 0; r < act_t; ++r)
      memcpy(&Z.kernel[(size_t)r * Z.cols], &Y.kernel[(size_t)r * Y.cols], (size_t)Y.cols * 2);
    if (feats)
      for (uint32_t f = 0; f < in_dim; ++f)   // feature order [sin u, sin v, cos u, cos v] x E (NifModel.cpp:216)
        memcpy(&Z.kernel[(size_t)(act_p + (f / E) * Ep + (f % E)) * Z.cols], &Y.kernel[(size_t)(act_t + f) * Y.cols], (size_t)Y.cols * 2);
    if (!Y.bias.empty()) {
      Z.bias.assign(Z.cols, 0);
      memcpy(Z.bias.data(), Y.bias.data(), (size_t)Y.cols * 2);
    }
    prev = Y.cols;
  }
  return PT_OK;
}

// ---- NIF weight packing -------------------------------------------------------------------
// Piece (l, j, s): the A operand of one v_mfma_f32_32x32x16_f16: lane (r = lane & 31, hh = lane >> 5)
// holds W^T[32 j + r][k(hh, 0..7)], where k() is the k-step's map onto rows of the Keras kernel:
//  * activation k-step s (from a previous accumulator tile t = s / 2, half s % 2):
//      k = 32 t + 16 (s % 2) + 8 (e >> 2) + 4 hh + (e & 3)     (accumulator-as-operand order)
//  * input k-step s' (Fourier features, NifModel.cpp:216 order [sin u, sin v, cos u, cos v]):
//      k = base + (e < 4 ? 0 : 2E) + hh E + 4 s' + (e & 3), base = H for a concat layer, else 0
int pack_nif(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<uint16_t>& wpack,
             std::vector<uint16_t>& bpack, ptd::NifParams& N) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E % 4) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be a multiple of 4");
  const uint32_t in_dim = 4 * E, H = L[0].cols;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (H % 32) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden size must be a multiple of 32");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  memset(&N, 0, sizeof(N));
  N.n_layers = n;
  uint32_t piece = 0, btile = 0;
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    const bool head = (l == n - 1);
    if (!head && Y.cols != H) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "all hidden layers must have the same width");
    bool concat = false;
    uint32_t act_steps = 0;
    if (l == 0) {
      if (Y.rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "bad first layer shape");
    } else if (Y.rows == H) {
      act_steps = H / 16;
    } else if (Y.rows == H + in_dim) {  // NifModel.cpp:305-308: x = concat(x, input)
      act_steps = H / 16;
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer input width is neither hidden nor hidden+features");
    }
    const uint32_t in_steps = (l == 0 || concat) ? E / 4 : 0;
    const uint32_t ksteps = act_steps + in_steps;
    const uint32_t ntiles = (Y.cols + 31) / 32;
    N.piece_base[l] = piece;
    N.bias_base[l] = btile;
    if (concat) N.concat_mask |= 1u << l;
    if (Y.relu) N.relu_mask |= 1u << l;
    if (!Y.bias.empty()) N.bias_mask |= 1u << l;
    wpack.resize((size_t)(piece + ntiles * ksteps) * 512, 0);
    bpack.resize((size_t)(btile + ntiles) * 32, 0);
    for (uint32_t j = 0; j < ntiles; ++j) {
      for (uint32_t s = 0; s < ksteps; ++s) {
        uint16_t* dst = &wpack[(size_t)(piece + j * ksteps + s) * 512];
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t r = lane & 31, hh = lane >> 5, col = 32 * j + r;
          for (uint32_t e = 0; e < 8; ++e) {
            uint32_t k;
            if (s < act_steps) {
              k = 32 * (s / 2) + 16 * (s % 2) + 8 * (e >> 2) + 4 * hh + (e & 3);
            } else {
              const uint32_t sp = s - act_steps;
              k = (concat ? H : 0) + (e < 4 ? 0 : 2 * E) + hh * E + 4 * sp + (e & 3);
            }
            dst[lane * 8 + e] = (col < Y.cols) ? Y.kernel[(size_t)k * Y.cols + col] : (uint16_t)0;
          }
        }
      }
      // bias of n-tile j in accumulator order: lane half hh, register i -> row (i&3) + 8 (i>>2) + 4 hh
      for (uint32_t hh = 0; hh < 2; ++hh)
        for (uint32_t i = 0; i < 16; ++i) {
          const uint32_t col = 32 * j + (i & 3) + 8 * (i >> 2) + 4 * hh;
          bpack[(size_t)(btile + j) * 32 + hh * 16 + i] = (!Y.bias.empty() && col < Y.cols) ? Y.bias[col] : (uint16_t)0;
        }
    }
    piece += ntiles * ksteps;
    btile += ntiles;
  }
  wpack.resize(wpack.size() + 512, 0);   // the paired loaders may copy (never use) one piece past the last
  return PT_OK;
}

// Wide networks (pt_nif_gemm.h), v_mfma_f32_16x16x32_f16.  Piece (l, s, f): the A operand of k-step s (32 inputs) and
// feature tile f (16 outputs) of layer l; lane (r = lane & 15, q = lane >> 4) holds W^T[16 f + r][k(q, 0..7)] with
//  * activation k-step s:  k = 32 s + (e < 4 ? 4 q + e : 16 + 4 q + (e - 4))          (accumulator-as-operand order)
//  * input k-step s':      coordinate cd = q & 1, frequency f' = 4 (2 s' + (q >> 1)) + (e & 3);
//                          k = base + (e < 4 ? 0 : 2E) + cd E + f', or a zero weight where f' >= E (padding slots)
// Pieces of a layer are ordered [s][f] (the two feature tiles a wave loads per stage are adjacent).  The head is one
// 16-row tile (rows 0..2), activation k-steps only: its feature inputs, if any, go to `head_in` as plain floats
// [sin u, sin v, cos u, cos v][E] x (B, G, R, -) for nifg16_finish_kernel.  Bias of a 32-feature group: [q][8]:
// e < 4 -> feature 32 j + 4 q + e, e >= 4 -> 32 j + 16 + 4 q + (e - 4).
int pack_nif_g16(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<uint16_t>& wpack,
                 std::vector<uint16_t>& bpack, ptd::NifParams& N, std::vector<float>& head_in, float head_bias[3],
                 uint32_t& head_piece_base) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E % 4) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be a multiple of 4");
  const uint32_t in_dim = 4 * E, H = L[0].cols;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (H % 256) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer-by-layer NIF path needs a hidden width that is a multiple of 256");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  memset(&N, 0, sizeof(N));
  N.n_layers = n;
  const uint32_t in_steps_all = (E / 4 + 1) / 2;
  uint32_t piece = 0, btile = 0;
  head_in.clear();
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    const bool head = (l == n - 1);
    if (!head && Y.cols != H) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "all hidden layers must have the same width");
    bool concat = false;
    uint32_t act_steps = 0;
    if (l == 0) {
      if (Y.rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "bad first layer shape");
    } else if (Y.rows == H) {
      act_steps = H / 32;
    } else if (Y.rows == H + in_dim) {  // NifModel.cpp:305-308: x = concat(x, input)
      act_steps = H / 32;
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer input width is neither hidden nor hidden+features");
    }
    const uint32_t in_steps = (!head && (l == 0 || concat)) ? in_steps_all : 0;
    const uint32_t ksteps = act_steps + in_steps;
    const uint32_t nf16 = head ? 1u : H / 16;
    N.piece_base[l] = piece;
    N.bias_base[l] = btile;
    if (concat) N.concat_mask |= 1u << l;
    if (Y.relu) N.relu_mask |= 1u << l;
    if (!Y.bias.empty()) N.bias_mask |= 1u << l;
    wpack.resize((size_t)(piece + ksteps * nf16) * 512, 0);
    for (uint32_t s = 0; s < ksteps; ++s)
      for (uint32_t f = 0; f < nf16; ++f) {
        uint16_t* dst = &wpack[(size_t)(piece + s * nf16 + f) * 512];
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t r = lane & 15, q = lane >> 4, col = 16 * f + r;
          for (uint32_t e = 0; e < 8; ++e) {
            uint32_t k;
            bool zero = col >= Y.cols;
            if (s < act_steps) {
              k = 32 * s + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
            } else {
              const uint32_t sp = s - act_steps, cd = q & 1, fr = 4 * (2 * sp + (q >> 1)) + (e & 3);
              if (fr >= E) zero = true;
              k = (concat ? H : 0) + (e < 4 ? 0 : 2 * E) + cd * E + fr;
            }
            dst[lane * 8 + e] = zero ? (uint16_t)0 : Y.kernel[(size_t)k * Y.cols + col];
          }
        }
      }
    if (head) {
      head_piece_base = piece;
      for (int o = 0; o < 3; ++o) head_bias[o] = Y.bias.empty() ? 0.f : host_h2f(Y.bias[o]);
      if (concat) {
        head_in.assign((size_t)in_dim * 4, 0.f);
        for (uint32_t f = 0; f < in_dim; ++f)
          for (int o = 0; o < 3; ++o) head_in[(size_t)f * 4 + o] = host_h2f(Y.kernel[(size_t)(H + f) * 3 + o]);
      }
    } else {
      const uint32_t nj = H / 32;
      bpack.resize((size_t)(btile + nj) * 32, 0);
      for (uint32_t j = 0; j < nj; ++j)
        for (uint32_t q = 0; q < 4; ++q)
          for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t col = 32 * j + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
            bpack[(size_t)(btile + j) * 32 + q * 8 + e] = Y.bias.empty() ? (uint16_t)0 : Y.bias[col];
          }
      btile += nj;
    }
    piece += ksteps * nf16;
  }
  bpack.resize(bpack.size() + 32, 0);
  return PT_OK;
}

#ifdef PTMI_DIAG_BUILD
// The same network packed for nif_kernel_v4 (v_mfma_f32_16x16x32_f16, pt_nif16.h).  Piece (l, j, s, ft): lane
// (r = lane & 15, qg = lane >> 4) holds W^T[32 j + 16 ft + r][k(qg, 0..7)] with
//  * activation k-step s:  k = 32 s + (e < 4 ? 4 qg + e : 16 + 4 qg + (e - 4))      (accumulator-as-operand order)
//  * input k-step s':      coordinate cd = qg & 1, frequency f = 4 (2 s' + (qg >> 1)) + (e & 3);
//                          k = base + (e < 4 ? 0 : 2E) + cd E + f, or a zero weight where f >= E (padding slots)
// Pieces of a layer are ordered (j, s, ft); the head has one 16-row tile, ordered (s).  Bias of a 32-feature tile:
// [qg][8]: e < 4 -> feature 32 j + 4 qg + e, e >= 4 -> 32 j + 16 + 4 qg + (e - 4).
int pack_nif16(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<uint16_t>& wpack,
               std::vector<uint16_t>& bpack, ptd::NifParams& N) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E % 4) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be a multiple of 4");
  const uint32_t in_dim = 4 * E, H = L[0].cols;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (H % 32) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden size must be a multiple of 32");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  memset(&N, 0, sizeof(N));
  N.n_layers = n;
  const uint32_t in_steps_all = (E / 4 + 1) / 2;
  uint32_t piece = 0, btile = 0;
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    const bool head = (l == n - 1);
    if (!head && Y.cols != H) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "all hidden layers must have the same width");
    bool concat = false;
    uint32_t act_steps = 0;
    if (l == 0) {
      if (Y.rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "bad first layer shape");
    } else if (Y.rows == H) {
      act_steps = H / 32;
    } else if (Y.rows == H + in_dim) {  // NifModel.cpp:305-308: x = concat(x, input)
      act_steps = H / 32;
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer input width is neither hidden nor hidden+features");
    }
    const uint32_t in_steps = (l == 0 || concat) ? in_steps_all : 0;
    const uint32_t ksteps = act_steps + in_steps;
    const uint32_t ntiles = head ? 1u : H / 32;
    const uint32_t nft = head ? 1u : 2u;
    N.piece_base[l] = piece;
    N.bias_base[l] = btile;
    if (concat) N.concat_mask |= 1u << l;
    if (Y.relu) N.relu_mask |= 1u << l;
    if (!Y.bias.empty()) N.bias_mask |= 1u << l;
    wpack.resize((size_t)(piece + ntiles * ksteps * nft) * 512, 0);
    bpack.resize((size_t)(btile + ntiles) * 32, 0);
    for (uint32_t j = 0; j < ntiles; ++j) {
      for (uint32_t s = 0; s < ksteps; ++s)
        for (uint32_t ft = 0; ft < nft; ++ft) {
          uint16_t* dst = &wpack[(size_t)(piece + (j * ksteps + s) * nft + ft) * 512];
          for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t r = lane & 15, qg = lane >> 4, col = 32 * j + 16 * ft + r;
            for (uint32_t e = 0; e < 8; ++e) {
              uint32_t k;
              bool zero = col >= Y.cols;
              if (s < act_steps) {
                k = 32 * s + (e < 4 ? 4 * qg + e : 16 + 4 * qg + (e - 4));
              } else {
                const uint32_t sp = s - act_steps, cd = qg & 1, f = 4 * (2 * sp + (qg >> 1)) + (e & 3);
                if (f >= E) zero = true;
                k = (concat ? H : 0) + (e < 4 ? 0 : 2 * E) + cd * E + f;
              }
              dst[lane * 8 + e] = zero ? (uint16_t)0 : Y.kernel[(size_t)k * Y.cols + col];
            }
          }
        }
      for (uint32_t qg = 0; qg < 4; ++qg)
        for (uint32_t e = 0; e < 8; ++e) {
          const uint32_t col = 32 * j + (e < 4 ? 4 * qg + e : 16 + 4 * qg + (e - 4));
          bpack[(size_t)(btile + j) * 32 + qg * 8 + e] = (!Y.bias.empty() && col < Y.cols) ? Y.bias[col] : (uint16_t)0;
        }
    }
    piece += ntiles * ksteps * nft;
    btile += ntiles;
  }
  return PT_OK;
}

#endif

#ifdef PTMI_DIAG_BUILD
template <int HID, int E, int WAVES, int TPS>
void launch_nif_v4(pt_handle h, const ptd::NifParams& N, int blocks) {
  using G = ptd::NifV4Geometry<HID, E, WAVES, TPS>;
  static std::atomic<unsigned long long> attr_set{0};   // one bit per device; the host app drives devices from threads
  if (!(attr_set.load(std::memory_order_relaxed) >> (h->cfg.device & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ptd::nif_kernel_v4<HID, E, WAVES, TPS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set.fetch_or(1ull << (h->cfg.device & 63), std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((ptd::nif_kernel_v4<HID, E, WAVES, TPS>), dim3(blocks), dim3(64 * WAVES), G::LDS_BYTES, h->stream, N);
}
#endif

// Dynamic-LDS opt-in of a kernel, once per device (one bit per device: the host app drives devices from threads).
int set_dynamic_lds(pt_handle h, const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  const unsigned long long bit = 1ull << (h->cfg.device & 63);
  if (done.load(std::memory_order_acquire) & bit) return PT_OK;
  PT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.fetch_or(bit, std::memory_order_release);
  return PT_OK;
}

template <int HID, int E, int NB, int WAVES>
int launch_nif_v2(pt_handle h, const ptd::NifParams& N, int blocks) {
  using G = ptd::NifV2Geometry<HID, E, WAVES>;
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nif_kernel_v2<HID, E, NB, WAVES>), G::LDS_BYTES, attr_set)) return rc;
  hipLaunchKernelGGL((ptd::nif_kernel_v2<HID, E, NB, WAVES>), dim3(blocks), dim3(64 * WAVES), G::LDS_BYTES, h->stream, N);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

template <int HID, int E, int WAVES, int TPS, int DIAG = 0>
int launch_nif_v3(pt_handle h, const ptd::NifParams& N, int blocks) {
  using G = ptd::NifV3Geometry<HID, E, WAVES, TPS>;
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nif_kernel_v3<HID, E, WAVES, TPS, DIAG>), G::LDS_BYTES, attr_set)) return rc;
  hipLaunchKernelGGL((ptd::nif_kernel_v3<HID, E, WAVES, TPS, DIAG>), dim3(blocks), dim3(64 * WAVES), G::LDS_BYTES, h->stream, N);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

#ifdef PTMI_DIAG_BUILD
// Timing-only ablations of the headline kernel (results are garbage): see nif_kernel_v3's DIAG bits.
template <int HID, int E>
bool launch_nif_diag(pt_handle h, const ptd::NifParams& N, int blocks) {
  const int diag = getenv("PTMI_NIF_DIAG") ? atoi(getenv("PTMI_NIF_DIAG")) : 0;   // read per launch: A/B rounds interleave in one process
  if constexpr (HID == 320 && E == 12) {
    switch (diag) {
      case 1: launch_nif_v3<HID, E, 8, 2, 1>(h, N, blocks); return true;
      case 2: launch_nif_v3<HID, E, 8, 2, 2>(h, N, blocks); return true;
      case 3: launch_nif_v3<HID, E, 8, 2, 3>(h, N, blocks); return true;
      case 4: launch_nif_v3<HID, E, 8, 2, 4>(h, N, blocks); return true;
      case 7: launch_nif_v3<HID, E, 8, 2, 7>(h, N, blocks); return true;
      case 15: launch_nif_v3<HID, E, 8, 2, 15>(h, N, blocks); return true;
      case 16: launch_nif_v3<HID, E, 8, 2, 16>(h, N, blocks); return true;
      case 32: launch_nif_v3<HID, E, 8, 2, 32>(h, N, blocks); return true;
      default: break;
    }
  }
  return false;
}
#endif

// Register-resident kernels, one instantiation per (padded hidden width, padded embedding): v3 keeps the bias tiles of
// at most 8 layers resident in LDS; deeper networks take the v2 ring (layer 0 resident).  A ring stage of v3 carries
// two output tiles where the tile count is even, else one.
template <int HID, int E>
int launch_nif_t(pt_handle h, const ptd::NifParams& N, int blocks) {
  constexpr int TPS = ((HID / 32) % 2 == 0) ? 2 : 1;
#ifdef PTMI_DIAG_BUILD
  if constexpr (HID == 320 && E == 12) {
    if (h->nif_m16) { launch_nif_v4<HID, E, 8, 2>(h, N, blocks); return PT_OK; }
    if (launch_nif_diag<HID, E>(h, N, blocks)) return PT_OK;
    // A/B switch of the profiling build: 1 = weights straight from L2, 2 = LDS ring with 4 waves x 64 samples
    const int variant = getenv("PTMI_NIF_VARIANT") ? atoi(getenv("PTMI_NIF_VARIANT")) : 0;
    if (variant == 1) { hipLaunchKernelGGL((ptd::nif_kernel<HID, E, 2>), dim3(blocks), dim3(256), 0, h->stream, N); return PT_OK; }
    if (variant == 2) return launch_nif_v2<HID, E, 2, 4>(h, N, blocks);
    if (variant == 3) return launch_nif_v2<HID, E, 1, 8>(h, N, blocks);
  }
#endif
  if (N.n_layers > (uint32_t)ptd::NifV3Geometry<HID, E, 8, TPS>::MAX_LAYERS) return launch_nif_v2<HID, E, 1, 8>(h, N, blocks);
  return launch_nif_v3<HID, E, 8, TPS>(h, N, blocks);
}

#ifdef PTMI_HEADLINE_ONLY
// Development build (seconds to compile): only the headline shape is instantiated.
template <int E>
int launch_nif_e(pt_handle h, const ptd::NifParams& N, int blocks) {
  if constexpr (E == 12) { if (h->nif_hidden == 320) return launch_nif_t<320, 12>(h, N, blocks); }
  return fail(h, PT_ERR_UNSUPPORTED_MODEL, "PTMI_HEADLINE_ONLY build: only hidden 320 / embedding 12 is instantiated");
}
#else
template <int E>
int launch_nif_e(pt_handle h, const ptd::NifParams& N, int blocks) {
  switch (h->nif_hidden) {
    case 32: return launch_nif_t<32, E>(h, N, blocks);
    case 64: return launch_nif_t<64, E>(h, N, blocks);
    case 96: return launch_nif_t<96, E>(h, N, blocks);
    case 128: return launch_nif_t<128, E>(h, N, blocks);
    case 160: return launch_nif_t<160, E>(h, N, blocks);
    case 192: return launch_nif_t<192, E>(h, N, blocks);
    case 224: return launch_nif_t<224, E>(h, N, blocks);
    case 256: return launch_nif_t<256, E>(h, N, blocks);
    case 288: return launch_nif_t<288, E>(h, N, blocks);
    case 320: return launch_nif_t<320, E>(h, N, blocks);
    default: break;
  }
  return fail(h, PT_ERR_UNSUPPORTED_MODEL, "no register-resident NIF kernel for hidden width " + std::to_string(h->nif_hidden));
}
#endif

#ifdef PTMI_DIAG_BUILD
template <int HID, int E>
int launch_nif_wide(pt_handle h, const ptd::NifParams& N, int blocks) {
  constexpr int lds = (HID / 16) * 2 * 1024 + (ptd::kMaxRegions + 1 + 256) * 4;
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nif_wide_kernel<HID, E>), lds, attr_set)) return rc;
  hipLaunchKernelGGL((ptd::nif_wide_kernel<HID, E>), dim3(blocks), dim3(256), lds, h->stream, N);
  return PT_OK;
}
#endif

#ifdef PTMI_DIAG_BUILD
// ---- profiling build: the round-2 32x32x16 layer path (A/B baseline)
template <int E>
void launch_nifg32_encode(pt_handle h, const ptd::NifParams& N, uint32_t tile0, uint32_t chunk) {
  hipLaunchKernelGGL((ptd::nifg_encode_kernel<E>), dim3((chunk + 3u) / 4u), dim3(256), 0, h->stream, N, h->d_tile_start, tile0, chunk,
                     h->d_gemm_feat);
}

// Wide networks, one launch per layer over chunks of the queue (pt_nif_gemm.h).  The number of queue tiles is only
// known on the device, so chunks are launched up to the queue's capacity and those past its end return at once.
int launch_nif_gemm32(pt_handle h, const ptd::NifParams& N) {
  const uint32_t H = (uint32_t)h->nif_hidden, KS = H / 16, IS = (uint32_t)h->nif_emb / 4, NT = H / 32, FB = NT / 8;
  const uint32_t n_layers = N.n_layers, chunk = h->gemm_chunk;
  if (!chunk) return fail(h, PT_ERR_NOT_READY, "wide-NIF buffers are not allocated");
  if (FB == 0 || NT % 8u) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer-by-layer NIF path needs a hidden width that is a multiple of 256");
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nifg_layer_kernel<0>), ptd::kGemmLdsBytes, attr_set)) return rc;
#ifdef PTMI_DIAG_BUILD
  // A/B switches of the profiling build, read per launch: PTMI_GEMM_KERNEL = v1 (round-1 interleaved kernel) | ld (ping-pong
  // + loader waves); PTMI_GEMM_DIAG = timing-only ablation bits of the selected kernel
  const char* gk = getenv("PTMI_GEMM_KERNEL");
  const int variant = !gk ? 0 : (!strcmp(gk, "v1") ? 1 : (!strcmp(gk, "ld") ? 2 : 0));
  const int gdiag = getenv("PTMI_GEMM_DIAG") ? atoi(getenv("PTMI_GEMM_DIAG")) : 0;
  auto launch_layer = [&](const ptd::NifGemmParams& G, uint32_t grid) -> int {
#define PT_LAYER(KERNEL, THREADS)                                                                                         \
    do {                                                                                                                  \
      PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                 ptd::kGemmLdsBytes));                                                                    \
      hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(THREADS), ptd::kGemmLdsBytes, h->stream, G);                            \
      return PT_OK;                                                                                                       \
    } while (0)
    if (variant == 2) PT_LAYER(ptd::nifg_layer_ld_kernel<0>, 768);
    if (variant == 1) switch (gdiag) {
      case 1: PT_LAYER(ptd::nifg_layer_v1_kernel<1>, 512); case 2: PT_LAYER(ptd::nifg_layer_v1_kernel<2>, 512);
      case 3: PT_LAYER(ptd::nifg_layer_v1_kernel<3>, 512); case 4: PT_LAYER(ptd::nifg_layer_v1_kernel<4>, 512);
      case 7: PT_LAYER(ptd::nifg_layer_v1_kernel<7>, 512); case 8: PT_LAYER(ptd::nifg_layer_v1_kernel<8>, 512);
      case 16: PT_LAYER(ptd::nifg_layer_v1_kernel<16>, 512); default: PT_LAYER(ptd::nifg_layer_v1_kernel<0>, 512);
    }
    switch (gdiag) {
      case 1: PT_LAYER(ptd::nifg_layer_kernel<1>, 512); case 2: PT_LAYER(ptd::nifg_layer_kernel<2>, 512);
      case 3: PT_LAYER(ptd::nifg_layer_kernel<3>, 512); case 4: PT_LAYER(ptd::nifg_layer_kernel<4>, 512);
      case 8: PT_LAYER(ptd::nifg_layer_kernel<8>, 512); case 16: PT_LAYER(ptd::nifg_layer_kernel<16>, 512);
      case 128: PT_LAYER(ptd::nifg_layer_kernel<128>, 512);   // two phases per stage (valid results)
      case 64: case 192: {   // stamped builds, four / two phases per stage (valid results): the stamps of the LAST layer launch are read by pt_diag_stamps
        ptd::NifGemmParams GS = G;
        GS.stamps = h->d_stamps;
        if (gdiag == 64) {
          PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ptd::nifg_layer_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, ptd::kGemmLdsBytes));
          hipLaunchKernelGGL(ptd::nifg_layer_kernel<64>, dim3(grid), dim3(512), ptd::kGemmLdsBytes, h->stream, GS);
        } else {
          PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ptd::nifg_layer_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, ptd::kGemmLdsBytes));
          hipLaunchKernelGGL(ptd::nifg_layer_kernel<192>, dim3(grid), dim3(512), ptd::kGemmLdsBytes, h->stream, GS);
        }
        return PT_OK;
      }
      default: break;
    }
#undef PT_LAYER
    hipLaunchKernelGGL(ptd::nifg_layer_kernel<0>, dim3(grid), dim3(512), ptd::kGemmLdsBytes, h->stream, G);
    return PT_OK;
  };
#else
  auto launch_layer = [&](const ptd::NifGemmParams& G, uint32_t grid) -> int {
    hipLaunchKernelGGL(ptd::nifg_layer_kernel<0>, dim3(grid), dim3(512), ptd::kGemmLdsBytes, h->stream, G);
    return PT_OK;
  };
#endif
  hipLaunchKernelGGL(ptd::nifg_scan_kernel, dim3(1), dim3(256), 0, h->stream, N.region_count, N.n_regions, h->d_tile_start);
  PT_HIP(hipGetLastError());
  const uint64_t max_tiles = (uint64_t)N.n_regions * ((N.region_cap + 31u) / 32u);
  uint32_t grid = ((uint32_t)h->n_cus / (8u * FB)) * 8u * FB;
  if (grid == 0) grid = 8u * FB;
  ptd::NifGemmParams G{};
  G.wpack = N.wpack;
  G.bpack = N.bpack;
  G.feat = h->d_gemm_feat;
  G.act_stride = KS;
  G.feat_stride = IS;
  G.total_tiles = h->d_tile_start + N.n_regions;
  G.chunk_tiles = chunk;
  for (uint64_t tile0 = 0; tile0 < max_tiles; tile0 += chunk) {
    G.tile0 = (uint32_t)tile0;
    switch (h->nif_emb) {
      case 4: launch_nifg32_encode<4>(h, N, G.tile0, chunk); break;
      case 8: launch_nifg32_encode<8>(h, N, G.tile0, chunk); break;
      case 12: launch_nifg32_encode<12>(h, N, G.tile0, chunk); break;
      case 16: launch_nifg32_encode<16>(h, N, G.tile0, chunk); break;
      default: return fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension");
    }
    PT_HIP(hipGetLastError());
    for (uint32_t l = 0; l + 1 < n_layers; ++l) {
      const bool concat = (N.concat_mask >> l) & 1u;
      G.piece_base = N.piece_base[l];
      G.bias_base = N.bias_base[l];
      G.ks_act = l ? KS : 0u;
      G.ks_in = (l == 0 || concat) ? IS : 0u;
      G.relu = (N.relu_mask >> l) & 1u;
      G.n_ftiles = NT;
      G.act_in = h->d_gemm_act[(l + 1u) & 1u];
      G.act_out = h->d_gemm_act[l & 1u];
      if (int rc = launch_layer(G, grid)) return rc;
      PT_HIP(hipGetLastError());
    }
    const uint32_t l = n_layers - 1;
    G.piece_base = N.piece_base[l];
    G.bias_base = N.bias_base[l];
    G.ks_act = KS;
    G.ks_in = ((N.concat_mask >> l) & 1u) ? IS : 0u;
    G.relu = (N.relu_mask >> l) & 1u;
    G.n_ftiles = 1;
    G.act_in = h->d_gemm_act[(l + 1u) & 1u];
    G.act_out = nullptr;
    hipLaunchKernelGGL(ptd::nifg_head_kernel, dim3((chunk + 15u) / 16u), dim3(256), 0, h->stream, N, G, h->d_tile_start);
    PT_HIP(hipGetLastError());
  }
  return PT_OK;
}

#endif

// Fork the chunk streams off the NIF stream / join them back (pt_context::chunk_stream).
static int chunk_streams_fork(pt_handle h, int sets) {
  if (sets < 2) return PT_OK;
  PT_HIP(hipEventRecord(h->chunk_fork, h->stream));
  for (int i = 0; i + 1 < sets; ++i) PT_HIP(hipStreamWaitEvent(h->chunk_stream[i], h->chunk_fork, 0));
  return PT_OK;
}
static int chunk_streams_join(pt_handle h, int sets, int rc) {   // also after a failed launch: whatever was queued ends before the NIF stream goes on
  for (int i = 0; i + 1 < sets; ++i) {
    const hipError_t e1 = hipEventRecord(h->chunk_join[i], h->chunk_stream[i]), e2 = hipStreamWaitEvent(h->stream, h->chunk_join[i], 0);
    if (rc == PT_OK && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(h, PT_ERR_HIP, "joining a chunk stream failed");
  }
  return rc;
}
static int chunk_sets_for(pt_handle h, uint64_t max_tiles, uint32_t chunk) {
  int sets = (int)std::min<uint64_t>((max_tiles + chunk - 1) / chunk, (uint64_t)h->chunk_sets);
#ifdef PTMI_DIAG_BUILD
  if (const char* e = getenv("PTMI_CHUNK_STREAMS")) sets = std::max(1, std::min(sets, atoi(e)));   // A/B of the profiling build
#endif
  return std::max(sets, 1);
}

// Wide networks, one launch per layer over chunks of the queue (pt_nif_gemm.h).  The number of queue tiles is only
// known on the device, so chunks are launched up to the queue's capacity and those past its end return at once.
template <int E>
void launch_nifg16_encode(pt_handle h, hipStream_t st, const ptd::NifParams& N, uint32_t tile0, uint32_t chunk, uint4* feat) {
  hipLaunchKernelGGL((ptd::nifg16_encode_kernel<E>), dim3((chunk + 3u) / 4u), dim3(256), 0, st, N, h->d_tile_start, tile0, chunk, feat);
}

template <int FUSE>
int launch_nifg16_layer(pt_handle h, hipStream_t st, const ptd::NifGemmParams& G, uint32_t grid) {
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nifg16_layer_kernel<FUSE, 0>), ptd::kGemmLdsBytes, attr_set)) return rc;
#ifdef PTMI_DIAG_BUILD
  // timing-only ablations / clock stamps of the profiling build, read per launch (PTMI_GEMM_DIAG: see the kernel's DIAG bits)
  const int gdiag = getenv("PTMI_GEMM_DIAG") ? atoi(getenv("PTMI_GEMM_DIAG")) : 0;
#define PT_LAYER16(D)                                                                                                          \
  case D: {                                                                                                                    \
    PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ptd::nifg16_layer_kernel<FUSE, D>),                               \
                               hipFuncAttributeMaxDynamicSharedMemorySize, ptd::kGemmLdsBytes));                               \
    ptd::NifGemmParams GS = G;                                                                                                 \
    GS.stamps = h->d_stamps;                                                                                                   \
    hipLaunchKernelGGL((ptd::nifg16_layer_kernel<FUSE, D>), dim3(grid), dim3(512), ptd::kGemmLdsBytes, st, GS);                 \
    PT_HIP(hipGetLastError());                                                                                                 \
    return PT_OK;                                                                                                              \
  }
  switch (gdiag) { PT_LAYER16(1) PT_LAYER16(2) PT_LAYER16(3) PT_LAYER16(8) PT_LAYER16(32) PT_LAYER16(64) PT_LAYER16(128) PT_LAYER16(256)
                   PT_LAYER16(16) PT_LAYER16(33) PT_LAYER16(40) PT_LAYER16(48) PT_LAYER16(160) PT_LAYER16(512) PT_LAYER16(544) PT_LAYER16(1024) PT_LAYER16(2048) PT_LAYER16(1056) PT_LAYER16(4096) PT_LAYER16(5120) default: break; }
#undef PT_LAYER16
#endif
  hipLaunchKernelGGL((ptd::nifg16_layer_kernel<FUSE, 0>), dim3(grid), dim3(512), ptd::kGemmLdsBytes, st, G);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int launch_nif_gemm(pt_handle h, const ptd::NifParams& N) {
  const uint32_t H = (uint32_t)h->nif_hidden, KS = H / 32, IS = ((uint32_t)h->nif_emb / 4 + 1) / 2, NT = H / 32, FB = NT / 8;
  const uint32_t n_layers = N.n_layers, chunk = h->gemm_chunk;
  if (!chunk) return fail(h, PT_ERR_NOT_READY, "wide-NIF buffers are not allocated");
  if (FB == 0 || NT % 8u) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer-by-layer NIF path needs a hidden width that is a multiple of 256");
  hipLaunchKernelGGL(ptd::nifg_scan_kernel, dim3(1), dim3(256), 0, h->stream, N.region_count, N.n_regions, h->d_tile_start);
  PT_HIP(hipGetLastError());
  const uint64_t max_tiles = (uint64_t)N.n_regions * ((N.region_cap + 31u) / 32u);
  uint32_t grid = ((uint32_t)h->n_cus / (8u * FB)) * 8u * FB;
  if (grid == 0) grid = 8u * FB;
  ptd::NifGemmParams G{};
  G.wpack = N.wpack;
  G.bpack = N.bpack;
  G.act_stride = KS;
  G.feat_stride = IS;
  G.total_tiles = h->d_tile_start + N.n_regions;
  G.chunk_tiles = chunk;
  G.n_ftiles = NT;
  G.head_piece_base = h->head_piece_base;
  G.partial_stride = chunk * 32u;
  const uint32_t lh = n_layers - 1;
  ptd::NifHeadParams Hd{};
  Hd.slices = 2u * FB;
  Hd.partial_stride = chunk * 32u;
  Hd.in_weights = ((N.concat_mask >> lh) & 1u) ? h->d_head_in : nullptr;
  Hd.n_in = (uint32_t)h->nif_emb;
  Hd.bias0 = h->head_bias[0]; Hd.bias1 = h->head_bias[1]; Hd.bias2 = h->head_bias[2];
  Hd.relu = (N.relu_mask >> lh) & 1u;
  Hd.chunk_tiles = chunk;
  // chunks alternate between the NIF stream and chunk_stream (each with its own buffer set): see pt_context::chunk_stream
  const int sets = chunk_sets_for(h, max_tiles, chunk);
  if (int frc = chunk_streams_fork(h, sets)) return frc;
  int rc = PT_OK;
  uint32_t set = 0;
  for (uint64_t tile0 = 0; tile0 < max_tiles && rc == PT_OK; tile0 += chunk, set = (set + 1u) % (uint32_t)sets) {
    hipStream_t st = set ? h->chunk_stream[set - 1u] : h->stream;
    uint4* const act[2] = {h->d_gemm_act[0] + set * h->gemm_act_set, h->d_gemm_act[1] + set * h->gemm_act_set};
    uint4* const feat = h->d_gemm_feat + set * h->gemm_feat_set;
    float4* const partial = h->d_head_partial + set * h->head_partial_set;
    G.tile0 = (uint32_t)tile0;
    G.feat = feat;
    switch (h->nif_emb) {
      case 4: launch_nifg16_encode<4>(h, st, N, G.tile0, chunk, feat); break;
      case 8: launch_nifg16_encode<8>(h, st, N, G.tile0, chunk, feat); break;
      case 12: launch_nifg16_encode<12>(h, st, N, G.tile0, chunk, feat); break;
      case 16: launch_nifg16_encode<16>(h, st, N, G.tile0, chunk, feat); break;
      default: rc = fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension"); continue;
    }
    if (hipGetLastError() != hipSuccess) { rc = fail(h, PT_ERR_HIP, "wide-NIF encode launch failed"); continue; }
    for (uint32_t l = 0; l + 1 < n_layers && rc == PT_OK; ++l) {
      const bool concat = (N.concat_mask >> l) & 1u;
      const bool last = l + 2 == n_layers;   // the head rides in this layer's epilogue
      G.piece_base = N.piece_base[l];
      G.bias_base = N.bias_base[l];
      G.ks_act = l ? KS : 0u;
      G.ks_in = (l == 0 || concat) ? IS : 0u;
      G.relu = (N.relu_mask >> l) & 1u;
      G.act_in = act[(l + 1u) & 1u];
      G.act_out = last ? nullptr : act[l & 1u];
      G.head_partial = last ? partial : nullptr;
      rc = last ? launch_nifg16_layer<1>(h, st, G, grid) : launch_nifg16_layer<0>(h, st, G, grid);
    }
    if (rc) continue;
    Hd.tile0 = G.tile0;
    Hd.partial = partial;
    hipLaunchKernelGGL(ptd::nifg16_finish_kernel, dim3((chunk + 7u) / 8u), dim3(256), 0, st, N, Hd, h->d_tile_start);
    if (hipGetLastError() != hipSuccess) rc = fail(h, PT_ERR_HIP, "wide-NIF finish launch failed");
  }
  return chunk_streams_join(h, sets, rc);
}

// ---- float32 models (pt_nif_f32.h) -----------------------------------------------------------------------------
// Shapes as normalize_nif: hidden widths padded to a common multiple of 32 (zero weights, zero bias), the Fourier features
// to 4 x Ep with Ep = E rounded up to a multiple of 4 (zero rows for the padding slots, zero features).
struct HostLayerF32 { uint32_t rows, cols; std::vector<float> kernel, bias; bool has_bias, relu; };

int pack_nif_f32(pt_handle h, const std::vector<HostLayerF32>& L, uint32_t E, std::vector<float>& blob,
                 std::vector<pt_context::F32Layer>& out, uint32_t& Hp_out, uint32_t& Ep_out) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E > 16) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be in 1..16");
  const uint32_t in_dim = 4 * E, Ep = (E + 3u) / 4u * 4u, in_p = 4 * Ep;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  uint32_t widest = 0;
  for (uint32_t l = 0; l + 1 < n; ++l) widest = std::max(widest, L[l].cols);
  const uint32_t Hp = (widest + 31u) / 32u * 32u;
  if (Hp > kMaxGemmHidden) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden layers wider than 2048 are not supported");
  out.clear();
  blob.clear();
  uint32_t prev = 0;
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayerF32& Y = L[l];
    const bool head = l + 1 == n;
    bool concat = false;
    if (l == 0) {
    } else if (Y.rows == prev) {
    } else if (Y.rows == prev + in_dim) {   // NifModel.cpp:305-308
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer " + std::to_string(l) + ": input width " + std::to_string(Y.rows) +
                                                   " is neither the previous layer's width nor that plus the 4*embedding features");
    }
    pt_context::F32Layer F{};
    F.k_act = l ? Hp : 0u;
    F.k_in = (l == 0 || concat) ? in_p : 0u;
    F.ldw = head ? 4u : Hp;
    F.relu = Y.relu;
    F.w_off = blob.size();
    blob.resize(blob.size() + (size_t)(F.k_act + F.k_in) * F.ldw, 0.f);
    float* W = &blob[F.w_off];
    const uint32_t act_t = l ? prev : 0u;
    for (uint32_t r = 0; r < act_t; ++r)
      for (uint32_t c = 0; c < Y.cols; ++c) W[(size_t)r * F.ldw + c] = Y.kernel[(size_t)r * Y.cols + c];
    if (F.k_in)
      for (uint32_t f = 0; f < in_dim; ++f)   // feature order [sin u, sin v, cos u, cos v] x E (NifModel.cpp:216)
        for (uint32_t c = 0; c < Y.cols; ++c)
          W[(size_t)(F.k_act + (f / E) * Ep + (f % E)) * F.ldw + c] = Y.kernel[(size_t)(act_t + f) * Y.cols + c];
    F.b_off = blob.size();
    blob.resize(blob.size() + F.ldw, 0.f);
    if (Y.has_bias) for (uint32_t c = 0; c < Y.cols; ++c) blob[F.b_off + c] = Y.bias[c];
    blob.resize((blob.size() + 3) / 4 * 4, 0.f);   // keep every kernel 16-byte aligned (the head reads float4 rows)
    out.push_back(F);
    prev = Y.cols;
  }
  Hp_out = Hp;
  Ep_out = Ep;
  return PT_OK;
}

template <int E>
void launch_nif32_encode(pt_handle h, hipStream_t st, const ptd::NifParams& N, uint32_t tile0, uint32_t chunk, float* feat) {
  hipLaunchKernelGGL((ptd::nif32_encode_kernel<E>), dim3((chunk + 3u) / 4u), dim3(256), 0, st, N, h->d_tile_start, tile0, chunk, feat);
}

int launch_nif_f32(pt_handle h, const ptd::NifParams& N) {
  const uint32_t chunk = h->f32_chunk, n_layers = (uint32_t)h->f32_layers.size();
  if (!chunk) return fail(h, PT_ERR_NOT_READY, "float32 NIF buffers are not allocated");
  hipLaunchKernelGGL(ptd::nifg_scan_kernel, dim3(1), dim3(256), 0, h->stream, N.region_count, N.n_regions, h->d_tile_start);
  PT_HIP(hipGetLastError());
  const uint64_t max_tiles = (uint64_t)N.n_regions * ((N.region_cap + 31u) / 32u);
  // chunks alternate between the NIF stream and chunk_stream (each with its own buffer set): see pt_context::chunk_stream
  const int sets = chunk_sets_for(h, max_tiles, chunk);
  if (int frc = chunk_streams_fork(h, sets)) return frc;
  int rc = PT_OK;
  uint32_t set = 0;
  for (uint64_t tile0 = 0; tile0 < max_tiles && rc == PT_OK; tile0 += chunk, set = (set + 1u) % (uint32_t)sets) {
    hipStream_t st = set ? h->chunk_stream[set - 1u] : h->stream;
    float* const act[2] = {h->d_f32_act[0] + set * h->f32_act_set, h->d_f32_act[1] + set * h->f32_act_set};
    float* const feat = h->d_f32_feat + set * h->f32_feat_set;
    switch (h->nif_emb) {
      case 4: launch_nif32_encode<4>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      case 8: launch_nif32_encode<8>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      case 12: launch_nif32_encode<12>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      case 16: launch_nif32_encode<16>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      default: rc = fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension"); continue;
    }
    if (hipGetLastError() != hipSuccess) { rc = fail(h, PT_ERR_HIP, "float32 NIF encode launch failed"); continue; }
    for (uint32_t l = 0; l < n_layers && rc == PT_OK; ++l) {
      const pt_context::F32Layer& F = h->f32_layers[l];
      const float* in = act[(l + 1u) & 1u];
      if (l + 1 < n_layers) {
        ptd::NifF32Params G{};
        G.w = h->d_f32_weights + F.w_off; G.bias = h->d_f32_weights + F.b_off;
        G.ldw = F.ldw; G.k_act = F.k_act; G.k_in = F.k_in; G.relu = F.relu;
        G.act_in = in; G.feat = feat; G.act_out = act[l & 1u];
        G.lda = h->f32_lda; G.ldf = h->f32_ldf;
        G.total_tiles = h->d_tile_start + N.n_regions; G.tile0 = (uint32_t)tile0; G.chunk_tiles = chunk;
        const uint32_t blocks = chunk / 8u * ((F.ldw + 63u) / 64u);  // one 256-sample x 64-feature block per workgroup
        hipLaunchKernelGGL(ptd::nif32_layer_kernel, dim3((blocks + 7u) / 8u * 8u), dim3(256), 0, st, G);
      } else {
        ptd::NifF32Head Hd{};
        Hd.w = h->d_f32_weights + F.w_off;
        Hd.k_act = F.k_act; Hd.k_in = F.k_in; Hd.relu = F.relu;
        Hd.bias0 = h->head_bias[0]; Hd.bias1 = h->head_bias[1]; Hd.bias2 = h->head_bias[2];
        Hd.act_in = in; Hd.feat = feat; Hd.lda = h->f32_lda; Hd.ldf = h->f32_ldf;
        Hd.tile0 = (uint32_t)tile0; Hd.chunk_tiles = chunk;
        hipLaunchKernelGGL(ptd::nif32_head_kernel, dim3((chunk + 7u) / 8u), dim3(256), 0, st, N, Hd, h->d_tile_start);
      }
      if (hipGetLastError() != hipSuccess) rc = fail(h, PT_ERR_HIP, "float32 NIF layer launch failed");
    }
  }
  return chunk_streams_join(h, sets, rc);
}

int launch_nif(pt_handle h, const ptd::NifParams& N, int blocks) {
  if (h->nif_f32) return launch_nif_f32(h, N);
  if (h->nif_gemm) {
#ifdef PTMI_DIAG_BUILD
    // A/B switch of the profiling build: the fused 64-sample kernel with activations in LDS
    static const bool fused = getenv("PTMI_NIF_WIDE") && !strcmp(getenv("PTMI_NIF_WIDE"), "fused");
    if (fused && h->nif_emb == 12 && h->nif_gemm32) {
      if (h->nif_hidden == 1024) return launch_nif_wide<1024, 12>(h, N, blocks);
      if (h->nif_hidden == 512) return launch_nif_wide<512, 12>(h, N, blocks);
    }
    if (h->nif_gemm32) return launch_nif_gemm32(h, N);
#endif
    return launch_nif_gemm(h, N);
  }
  switch (h->nif_emb) {
    case 4: return launch_nif_e<4>(h, N, blocks);
    case 8: return launch_nif_e<8>(h, N, blocks);
    case 12: return launch_nif_e<12>(h, N, blocks);
    case 16: return launch_nif_e<16>(h, N, blocks);
    default: break;
  }
  return fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension");
}

void free_batch_buffers(pt_handle h) {
  for (auto& B : h->bb) {
    (void)hipFree(B.q_u); (void)hipFree(B.q_v); (void)hipFree(B.q_tr); (void)hipFree(B.q_tg); (void)hipFree(B.q_tb);
    (void)hipFree(B.q_path); (void)hipFree(B.survivors); (void)hipFree(B.states); (void)hipFree(B.region_count); (void)hipFree(B.plen);
    (void)hipFree(B.rad_r); (void)hipFree(B.rad_g); (void)hipFree(B.rad_b);
    if (B.traced) (void)hipEventDestroy(B.traced);
    if (B.accumulated) (void)hipEventDestroy(B.accumulated);
    B = pt_context::BatchBuffers();
  }
}

hipEvent_t get_event(pt_handle h, size_t i) {
  while (h->events.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    h->events.push_back(e);
  }
  return h->events[i];
}

}  // namespace

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
#define PT_HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->error = std::string(#call) + ": " + hipGetErrorString(e_); return bail(PT_ERR_HIP); } } while (0)
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
  // auto: ~32 M paths per batch (2.5 GB of batch buffers at most -- small against 288 GB of HBM3E; larger batches
  // mean fewer launch tails: 10 instead of 43 NIF launches per 300-spp step bought 2 %)
  if (k == 0) { k = (uint32_t)((32u << 20) / n); if (k < 1) k = 1; if (k > 32) k = 32; }
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
  PT_HIPC(dev_alloc(&h->d_counters, 2));
  const TraceGrid g = trace_grid((uint32_t)h->batch_paths_cap);
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
  // profiling build: PTMI_SERIAL=1 runs the trace kernels on the NIF stream (no overlap) to measure interference
  if (getenv("PTMI_SERIAL")) { (void)hipStreamDestroy(h->trace_stream); h->trace_stream = h->stream; h->serial = true; }
#endif
#undef PT_HIPC
  *out = h;
  return PT_OK;
}

int pt_destroy(pt_handle h) {
  if (!h) return PT_OK;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  free_batch_buffers(h);
  (void)hipFree(h->d_records);
  (void)hipFree(h->acc.pix); (void)hipFree(h->acc.r); (void)hipFree(h->acc.g); (void)hipFree(h->acc.b); (void)hipFree(h->acc.count); (void)hipFree(h->acc.length);
  (void)hipFree(h->d_counters);
  (void)hipFree(h->d_wpack); (void)hipFree(h->d_bpack);
  (void)hipFree(h->d_gemm_act[0]); (void)hipFree(h->d_gemm_act[1]); (void)hipFree(h->d_gemm_feat); (void)hipFree(h->d_tile_start);
  (void)hipFree(h->d_stamps);
  (void)hipFree(h->d_head_partial); (void)hipFree(h->d_head_in);
  (void)hipFree(h->d_f32_weights); (void)hipFree(h->d_f32_act[0]); (void)hipFree(h->d_f32_act[1]); (void)hipFree(h->d_f32_feat);
  (void)hipFree(h->d_scratch);
  (void)hipFree(h->d_hdr_stage); (void)hipFree(h->d_hdr_gather); (void)hipFree(h->d_film);
  (void)hipFree(h->d_slot_check);
  (void)hipFree(h->tiles.cost); (void)hipFree(h->d_tile_tmp);
  if (h->comm) (void)ncclCommDestroy(h->comm);
  for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
  if (h->trace_stream && !h->serial) { (void)hipStreamSynchronize(h->trace_stream); (void)hipStreamDestroy(h->trace_stream); }
  if (h->acc_stream) { (void)hipStreamSynchronize(h->acc_stream); (void)hipStreamDestroy(h->acc_stream); }
  for (int i = 0; i + 1 < pt_context::kChunkSets; ++i) {
    if (h->chunk_stream[i]) { (void)hipStreamSynchronize(h->chunk_stream[i]); (void)hipStreamDestroy(h->chunk_stream[i]); }
    if (h->chunk_join[i]) (void)hipEventDestroy(h->chunk_join[i]);
  }
  if (h->chunk_fork) (void)hipEventDestroy(h->chunk_fork);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PT_OK;
}

// A model all of whose layers are float32: the float path (pt_nif_f32.h).
static int upload_nif_f32(pt_handle h, const pt_layer* layers, uint32_t n_layers, uint32_t embedding_dim, float max,
                          const float mean[3], int32_t log_tonemap) {
  std::vector<HostLayerF32> L(n_layers);
  uint64_t flops = 0;
  for (uint32_t l = 0; l < n_layers; ++l) {
    L[l].rows = layers[l].rows; L[l].cols = layers[l].cols; L[l].relu = layers[l].relu != 0; L[l].has_bias = layers[l].bias != nullptr;
    const float* kp = static_cast<const float*>(layers[l].kernel);
    L[l].kernel.assign(kp, kp + (size_t)L[l].rows * L[l].cols);
    if (layers[l].bias) { const float* bp = static_cast<const float*>(layers[l].bias); L[l].bias.assign(bp, bp + L[l].cols); }
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
  h->nif_gemm32 = false;
  h->nif_m16 = false;
  h->nif_flops = flops;
  h->nif_valid = true;
  h->env_const = false;
  return PT_OK;
}

int pt_upload_nif(pt_handle h, const pt_layer* layers, uint32_t n_layers, uint32_t embedding_dim, float max,
                  const float mean[3], int32_t log_tonemap) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!layers || !mean || n_layers == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "null NIF arguments");
  std::vector<HostLayer> L(n_layers);
  uint64_t flops = 0;
  bool all_f32 = true;
  for (uint32_t l = 0; l < n_layers; ++l) {
    if (layers[l].dtype != PT_DTYPE_F16 && layers[l].dtype != PT_DTYPE_F32)
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF weights must be float16 or float32");
    if (!layers[l].kernel || layers[l].rows == 0 || layers[l].cols == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "empty layer kernel");
    all_f32 = all_f32 && layers[l].dtype == PT_DTYPE_F32;
  }
  if (all_f32) return upload_nif_f32(h, layers, n_layers, embedding_dim, max, mean, log_tonemap);
  for (uint32_t l = 0; l < n_layers; ++l) {
    L[l].rows = layers[l].rows;
    L[l].cols = layers[l].cols;
    L[l].relu = layers[l].relu != 0;
    const size_t count = (size_t)L[l].rows * L[l].cols;
    if (layers[l].dtype == PT_DTYPE_F16) {
      const uint16_t* kp = static_cast<const uint16_t*>(layers[l].kernel);
      L[l].kernel.assign(kp, kp + count);
      if (layers[l].bias) {
        const uint16_t* bp = static_cast<const uint16_t*>(layers[l].bias);
        L[l].bias.assign(bp, bp + L[l].cols);
      }
    } else {
      // a float32 layer inside a float16 model (the reference would run that one layer in float; a model all of whose
      // layers are float32 takes the float path, upload_nif_f32): rounded to binary16 (RNE) here -- DESIGN.md section 2
      const float* kp = static_cast<const float*>(layers[l].kernel);
      L[l].kernel.resize(count);
      for (size_t i = 0; i < count; ++i) L[l].kernel[i] = host_f2h(kp[i]);
      if (layers[l].bias) {
        const float* bp = static_cast<const float*>(layers[l].bias);
        L[l].bias.resize(L[l].cols);
        for (uint32_t i = 0; i < L[l].cols; ++i) L[l].bias[i] = host_f2h(bp[i]);
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
  bool m16 = false, gemm32 = false;
  std::vector<float> head_in;
  float head_bias[3] = {0, 0, 0};
  uint32_t head_piece_base = 0;
#ifdef PTMI_DIAG_BUILD
  // A/B switch of the profiling build: PTMI_GEMM_SHAPE=32 keeps a wide network on the round-2 32x32x16 layer kernels
  if (const char* k = getenv("PTMI_GEMM_SHAPE")) gemm32 = plan.gemm && atoi(k) == 32;
  // A/B switch of the profiling build: PTMI_NIF_KERNEL=v4 packs for the 16x16x32 kernel (pt_nif16.h).  Measured equal
  // to v3 in NIF time within 1-2 %, but its 230 VGPRs leave no room for the trace kernel's waves beside it, so
  // the step is 1.5 % slower end to end (profiles/r01_c_nif_ablation.txt); v3 stays the product kernel.
  if (const char* k = getenv("PTMI_NIF_KERNEL")) m16 = strcmp(k, "v4") == 0 && plan.Hp == 320 && plan.Ep == 12 && n_layers <= 8;
  rc = m16 ? pack_nif16(h, padded, plan.Ep, wpack, bpack, N)
       : (plan.gemm && !gemm32) ? pack_nif_g16(h, padded, plan.Ep, wpack, bpack, N, head_in, head_bias, head_piece_base)
                                : pack_nif(h, padded, plan.Ep, wpack, bpack, N);
#else
  rc = plan.gemm ? pack_nif_g16(h, padded, plan.Ep, wpack, bpack, N, head_in, head_bias, head_piece_base)
                 : pack_nif(h, padded, plan.Ep, wpack, bpack, N);
#endif
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
  h->nif_gemm32 = gemm32;
  h->nif_m16 = m16;
  h->nif_flops = flops;
  h->nif_valid = true;
  h->env_const = false;
  return PT_OK;
}

int pt_set_constant_env(pt_handle h, const float rgb[3]) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!rgb) return fail(h, PT_ERR_INVALID_ARGUMENT, "null rgb");
  h->env_const = true;
  memcpy(h->env_rgb, rgb, 12);
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
  if (n == 0) return PT_OK;
  static_assert(sizeof(pt_trace_record) == 20 && sizeof(ptd::TraceRecordDev) == 20, "TraceRecord wire format");
  PT_HIP(hipMemcpyAsync(h->d_records, work, n * sizeof(pt_trace_record), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(ptd::unpack_records_kernel, dim3(((uint32_t)n + 255) / 256), dim3(256), 0, h->stream, h->d_records,
                     (uint32_t)n, h->acc);
  PT_HIP(hipGetLastError());
  PT_HIP(hipStreamSynchronize(h->stream));  // host buffer is not touched after return
  return PT_OK;
}

// Enqueue the whole path_trace program on the three streams.  Returns at the first failing call; the caller drains the
// streams either way, so a failure in the middle of the batch loop never leaves kernels running on buffers the host
// is about to reuse or free.
struct StageSpan { size_t a, b; int kind; };   // event pair around one stage of one batch: 0 trace, 1 NIF, 2 accumulate

static int enqueue_path_trace(pt_handle h, std::vector<StageSpan>& spans, size_t& e_begin_i, size_t& e_end_i) {
  const uint32_t n = h->n_items;
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
  uint32_t done = 0, batch = 0;
  while (done < h->samples_per_step) {
    // The first batch is kept short when the step has several: its trace kernel is the only one with no NIF kernel to
    // hide under, so the sooner it ends the sooner the MFMA pipes start (per-pixel sums stay in iteration order).
    uint32_t iters = std::min(h->iters_per_batch, h->samples_per_step - done);
    if (batch == 0 && !h->env_const && h->samples_per_step > 2u * h->iters_per_batch) iters = std::min(iters, h->first_batch_iters);
    const uint32_t total = iters * n;
    const TraceGrid g = trace_grid(total);
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
    // A/B switch of the profiling build, read per launch: the round-2 one-phase kernel
    const char* tk = getenv("PTMI_TRACE_KERNEL");
    if (tk && !strcmp(tk, "v1"))
      hipLaunchKernelGGL(ptd::trace_kernel_v1, dim3(g.blocks), dim3(ptd::kTraceBlock), 0, h->trace_stream, P);
    else if (tk && !strcmp(tk, "r1"))
      hipLaunchKernelGGL(ptd::trace_kernel_refill<1>, dim3(g.blocks), dim3(ptd::kTraceBlock), 0, h->trace_stream, P);
    else if (tk && !strcmp(tk, "r4"))
      hipLaunchKernelGGL(ptd::trace_kernel_refill<4>, dim3(g.blocks), dim3(ptd::kTraceBlock), 0, h->trace_stream, P);
    else if (tk && !strcmp(tk, "r16"))
      hipLaunchKernelGGL(ptd::trace_kernel_refill<16>, dim3(g.blocks), dim3(ptd::kTraceBlock), 0, h->trace_stream, P);
    else if (tk && !strcmp(tk, "r24"))
      hipLaunchKernelGGL(ptd::trace_kernel_refill<24>, dim3(g.blocks), dim3(ptd::kTraceBlock), 0, h->trace_stream, P);
    else
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
      spans.push_back({ev + 2, ev + 3, 1});
      h->stats.nif_launches += 1;
    }
    PT_HIP(hipEventRecord(n1, h->stream));
    PT_HIP(hipStreamWaitEvent(h->acc_stream, n1, 0));
    PT_HIP(hipEventRecord(a0, h->acc_stream));
    hipLaunchKernelGGL(ptd::accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, h->acc_stream, n, iters, B.plen, B.rad_r,
                       B.rad_g, B.rad_b, h->acc, h->d_counters);
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
  e_end_i = ev;
  hipEvent_t e_end = get_event(h, ev++);
  if (!e_end) return fail(h, PT_ERR_HIP, "hipEventCreate failed");
  PT_HIP(hipEventRecord(e_end, h->stream));
  return PT_OK;
}

int pt_path_trace(pt_handle h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!h->settings_valid) return fail(h, PT_ERR_NOT_READY, "pt_set_render_settings has not been called");
  if (!h->env_const && !h->nif_valid) return fail(h, PT_ERR_NOT_READY, "no environment: call pt_upload_nif or pt_set_constant_env");
  PT_HIP(hipSetDevice(h->cfg.device));
  memset(&h->stats, 0, sizeof(h->stats));
  h->stats.nif_flops_per_sample = h->env_const ? 0 : h->nif_flops;
  h->stats.first_sample = h->sample_cursor;
  const uint32_t n = h->n_items;
  if (n == 0) return PT_OK;

  std::vector<StageSpan> spans;
  size_t e_begin_i = 0, e_end_i = 0;
  const int rc = enqueue_path_trace(h, spans, e_begin_i, e_end_i);
  // Drain all three streams whether or not the enqueue succeeded (a kernel fault surfaces here as well).
  hipError_t s3 = hipSuccess;   // the chunk streams (forked from and joined to the NIF stream) first: a failed enqueue leaves nothing behind
  for (hipStream_t cs : h->chunk_stream) { const hipError_t e = hipStreamSynchronize(cs); if (e != hipSuccess) s3 = e; }
  const hipError_t s0 = hipStreamSynchronize(h->stream), s1 = hipStreamSynchronize(h->trace_stream),
                   s2 = hipStreamSynchronize(h->acc_stream);
  if (rc) return rc;   // h->error names the failing call
  for (hipError_t e : {s0, s1, s2, s3})
    if (e != hipSuccess) return fail(h, PT_ERR_HIP, std::string("path_trace: ") + hipGetErrorString(e));
  h->sample_cursor += h->samples_per_step;

  unsigned long long counters[2] = {0, 0};
  PT_HIP(hipMemcpy(counters, h->d_counters, sizeof(counters), hipMemcpyDeviceToHost));
  h->stats.paths = (uint64_t)n * h->samples_per_step;
  h->stats.segments = counters[0];
  h->stats.escaped = counters[1];
  for (const StageSpan& s : spans) {
    float ms = 0.f;
    PT_HIP(hipEventElapsedTime(&ms, h->events[s.a], h->events[s.b]));
    if (s.kind == 0) h->stats.path_trace_ms += ms;
    else if (s.kind == 1) h->stats.nif_ms += ms;
    else h->stats.accumulate_ms += ms;
  }
  float total_ms = 0.f;
  PT_HIP(hipEventElapsedTime(&total_ms, h->events[e_begin_i], h->events[e_end_i]));
  h->stats.total_ms = total_ms;
  return PT_OK;
}

int pt_get_stats(pt_handle h, pt_stats* stats) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!stats) return fail(h, PT_ERR_INVALID_ARGUMENT, "null stats");
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
  if (stats) *stats = h->stats;
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

// ---- multi-GPU film hand-off over RCCL --------------------------------------------------------------------------
// The path shards over pixels with no exchange of ray data (reference: one NIF replica per IPU, "no inter-ipu exchange",
// PathTracerApp.cpp:205-252; results only meet on the host film, AccumulatedImage.cpp:59-74).  The one exchange step is
// this gather of HDR tiles to rank 0 at a save interval: every peer sends its tile straight to the root over its own
// xGMI link (grouped ncclSend / ncclRecv -- never a ring), 12 B per work item.
//
// No call in here can block for ever.  Communicators are NON-BLOCKING (ncclConfig_t::blocking = 0): every RCCL call
// returns at once, and its completion -- connection set-up with a peer included -- is polled with
// ncclCommGetAsyncError against the handle's deadline (pt_comm_set_timeout, default 120 s); the device side is polled
// with hipStreamQuery against the same deadline.  On expiry, on an asynchronous RCCL error, or when another thread asks
// (pt_comm_abort) the communicator is aborted (ncclCommAbort ends the kernels still waiting for a peer), the stream is
// drained, and the call returns PT_ERR_COMM; the handle then refuses further gathers until it is given a new communicator.
// Every step that can fail locally (argument checks, allocations, the export kernel) runs BEFORE a rank enters the
// exchange, so a rank that returns early never leaves its peers inside a collective it has half joined: they time out.

#define PT_NCCL(call)                                                                        \
  do {                                                                                       \
    ncclResult_t r_ = (call);                                                                \
    if (r_ != ncclSuccess && r_ != ncclInProgress) {                                         \
      h->error = std::string(#call) + ": " + ncclGetErrorString(r_);                         \
      return PT_ERR_COMM;                                                                    \
    }                                                                                        \
  } while (0)

using comm_clock = std::chrono::steady_clock;

static void comm_backoff(unsigned& spins) {
  if (++spins < 200) std::this_thread::yield();
  else std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 50 : 500));
}

// Abort the handle's communicator and leave the handle without one.  Kernels of this communicator still spinning on a
// peer see the abort flag and exit, so the stream can be drained afterwards.
static void comm_abort_now(pt_handle h) {
  if (h->comm) (void)ncclCommAbort(h->comm);
  h->comm = nullptr;
  h->comm_broken = true;
  h->comm_slot_agreed = 0;
  h->comm_abort_req.store(false);
}

static int comm_fail(pt_handle h, const std::string& why) {
  comm_abort_now(h);
  (void)hipStreamSynchronize(h->stream);   // nothing of the aborted exchange is left running on the caller's buffers
  h->error = why + " -- communicator aborted";
  return PT_ERR_COMM;
}

// Host side of a non-blocking RCCL call: wait until the communicator has left ncclInProgress.
static int comm_wait_host(pt_handle h, const char* what, comm_clock::time_point deadline) {
  unsigned spins = 0;
  for (;;) {
    ncclResult_t st = ncclSuccess;
    const ncclResult_t q = ncclCommGetAsyncError(h->comm, &st);
    if (q != ncclSuccess) return comm_fail(h, std::string(what) + ": ncclCommGetAsyncError: " + ncclGetErrorString(q));
    if (st == ncclSuccess) return PT_OK;
    if (st != ncclInProgress) return comm_fail(h, std::string(what) + ": " + ncclGetErrorString(st));
    if (h->comm_abort_req.load()) return comm_fail(h, std::string(what) + ": aborted by pt_comm_abort");
    if (comm_clock::now() > deadline)
      return comm_fail(h, std::string(what) + ": no progress within " + std::to_string(h->comm_timeout_ms) + " ms (a peer is missing or has failed)");
    comm_backoff(spins);
  }
}

// Device side: wait until everything queued on the handle's stream has finished.
static int comm_wait_stream(pt_handle h, const char* what, comm_clock::time_point deadline) {
  unsigned spins = 0;
  for (;;) {
    const hipError_t e = hipStreamQuery(h->stream);
    (void)hipGetLastError();   // hipErrorNotReady must not surface from a later hipGetLastError()
    if (e == hipSuccess) return PT_OK;
    if (e != hipErrorNotReady) {
      const std::string msg = std::string(what) + ": " + hipGetErrorString(e);
      comm_abort_now(h);
      h->error = msg;
      return PT_ERR_HIP;
    }
    if (h->comm) {
      ncclResult_t st = ncclSuccess;
      if (ncclCommGetAsyncError(h->comm, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress)
        return comm_fail(h, std::string(what) + ": " + ncclGetErrorString(st));
      if (h->comm_abort_req.load()) return comm_fail(h, std::string(what) + ": aborted by pt_comm_abort");
      if (comm_clock::now() > deadline)
        return comm_fail(h, std::string(what) + ": the exchange did not finish within " + std::to_string(h->comm_timeout_ms) + " ms (a peer is missing or has failed)");
    }
    comm_backoff(spins);
  }
}

static comm_clock::time_point comm_deadline(pt_handle h) {
  return comm_clock::now() + std::chrono::milliseconds(h->comm_timeout_ms);
}

int pt_comm_get_unique_id(void* id_out) {
  static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must match ncclUniqueId");
  if (!id_out) { g_create_error = "null id buffer"; return PT_ERR_INVALID_ARGUMENT; }
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return PT_ERR_COMM; }
  memcpy(id_out, &id, sizeof(id));
  return PT_OK;
}

int pt_comm_set_timeout(pt_handle h, uint32_t milliseconds) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (milliseconds == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "the communicator deadline must be > 0 ms");
  h->comm_timeout_ms = milliseconds;
  return PT_OK;
}

int pt_comm_abort(pt_handle h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  h->comm_abort_req.store(true);   // the only field another thread may touch; the owning thread's polling loop acts on it
  return PT_OK;
}

static int comm_local_buffers(pt_handle h) {
  if (!h->d_slot_check) PT_HIP(dev_alloc(&h->d_slot_check, 2));
  return PT_OK;
}

int pt_comm_init_rank(pt_handle h, const void* id_in, int rank, int world) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!id_in || world < 1 || rank < 0 || rank >= world) return fail(h, PT_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  if (h->comm) return fail(h, PT_ERR_INVALID_ARGUMENT, "the handle already has a communicator");
  PT_HIP(hipSetDevice(h->cfg.device));
  if (int rc = comm_local_buffers(h)) return rc;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  cfg.blocking = 0;
  h->comm_broken = false;
  h->comm_abort_req.store(false);
  h->comm_slot_agreed = 0;
  const auto deadline = comm_deadline(h);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = ncclCommInitRankConfig(&comm, world, id, rank, &cfg);
  if (r != ncclSuccess && r != ncclInProgress) {
    if (comm) (void)ncclCommAbort(comm);
    return fail(h, PT_ERR_COMM, std::string("ncclCommInitRankConfig: ") + ncclGetErrorString(r));
  }
  h->comm = comm;
  if (int rc = comm_wait_host(h, "communicator set-up", deadline)) return rc;   // a rank that never arrives ends here, not in a hang
  h->comm_rank = rank;
  h->comm_world = world;
  return PT_OK;
}

int pt_comm_init_all(pt_handle* handles, int n) {
  if (!handles || n < 1) { g_create_error = "bad communicator arguments"; return PT_ERR_INVALID_ARGUMENT; }
  pt_handle h = handles[0];
  if (!h) { g_create_error = "null handle"; return PT_ERR_INVALID_ARGUMENT; }
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!handles[i]) return fail(h, PT_ERR_INVALID_ARGUMENT, "null handle in the list");
    if (handles[i]->comm) return fail(h, PT_ERR_INVALID_ARGUMENT, "a handle already has a communicator");
    devs[i] = handles[i]->cfg.device;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) return fail(h, PT_ERR_INVALID_ARGUMENT, "RCCL needs one device per rank: two handles share device " + std::to_string(devs[i]));
  }
  for (int i = 0; i < n; ++i) {   // local, fallible steps first
    if (hipSetDevice(devs[i]) != hipSuccess) return fail(h, PT_ERR_HIP, "hipSetDevice failed for device " + std::to_string(devs[i]));
    if (int rc = comm_local_buffers(handles[i])) { h->error = handles[i]->error; return rc; }
  }
  ncclUniqueId id;
  PT_NCCL(ncclGetUniqueId(&id));
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  cfg.blocking = 0;
  std::vector<ncclComm_t> comms(n, nullptr);
  auto abort_all = [&]() { for (auto c : comms) if (c) (void)ncclCommAbort(c); };
  // one process, several devices: the rank-wise initialisations form one group
  PT_NCCL(ncclGroupStart());
  for (int i = 0; i < n; ++i) {
    ncclResult_t r = (hipSetDevice(devs[i]) == hipSuccess) ? ncclCommInitRankConfig(&comms[i], n, id, i, &cfg) : ncclUnhandledCudaError;
    if (r != ncclSuccess && r != ncclInProgress) {
      (void)ncclGroupEnd();
      abort_all();
      return fail(h, PT_ERR_COMM, std::string("ncclCommInitRankConfig: ") + ncclGetErrorString(r));
    }
  }
  {
    const ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess && r != ncclInProgress) { abort_all(); return fail(h, PT_ERR_COMM, std::string("ncclGroupEnd: ") + ncclGetErrorString(r)); }
  }
  const auto deadline = comm_deadline(h);
  unsigned spins = 0;
  for (int i = 0; i < n;) {
    ncclResult_t st = ncclSuccess;
    const ncclResult_t q = comms[i] ? ncclCommGetAsyncError(comms[i], &st) : ncclInternalError;
    if (q == ncclSuccess && st == ncclSuccess) { ++i; continue; }
    if (q != ncclSuccess || st != ncclInProgress || comm_clock::now() > deadline) {
      abort_all();
      return fail(h, PT_ERR_COMM, "communicator set-up of rank " + std::to_string(i) + " failed or timed out: " +
                                      ncclGetErrorString(q != ncclSuccess ? q : st));
    }
    comm_backoff(spins);
  }
  for (int i = 0; i < n; ++i) {
    handles[i]->comm = comms[i];
    handles[i]->comm_rank = i;
    handles[i]->comm_world = n;
    handles[i]->comm_broken = false;
    handles[i]->comm_abort_req.store(false);
    handles[i]->comm_slot_agreed = 0;
  }
  return PT_OK;
}

int pt_film_accumulate(pt_handle h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  PT_HIP(hipSetDevice(h->cfg.device));
  if (!h->d_film) {
    PT_HIP(dev_alloc(&h->d_film, (size_t)h->capacity * 3));
    PT_HIP(hipMemsetAsync(h->d_film, 0, (size_t)h->capacity * 12, h->stream));
  }
  if (h->n_items) {
    hipLaunchKernelGGL(ptd::film_accumulate_kernel, dim3((h->n_items + 255) / 256), dim3(256), 0, h->stream, h->n_items, h->acc, h->d_film, h->tiles);
    PT_HIP(hipGetLastError());
  }
  h->film_steps += 1;
  return PT_OK;
}

int pt_film_seed(pt_handle h, const float* host_bgr, size_t n) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (n != h->n_items || (!host_bgr && n)) return fail(h, PT_ERR_INVALID_ARGUMENT, "film seed must cover exactly the current work items");
  PT_HIP(hipSetDevice(h->cfg.device));
  if (!h->d_film) {
    PT_HIP(dev_alloc(&h->d_film, (size_t)h->capacity * 3));
    PT_HIP(hipMemsetAsync(h->d_film, 0, (size_t)h->capacity * 12, h->stream));
  }
  if (n) PT_HIP(hipMemcpyAsync(h->d_film, host_bgr, n * 12, hipMemcpyHostToDevice, h->stream));
  PT_HIP(hipStreamSynchronize(h->stream));   // host buffer is not touched after return
  return PT_OK;
}

int pt_tile_costs_enable(pt_handle h, uint32_t tile_w, uint32_t tile_h) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (tile_w == 0 || tile_h == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "tile size must be > 0");
  PT_HIP(hipSetDevice(h->cfg.device));
  const uint32_t tx = (h->cfg.width + tile_w - 1) / tile_w, ty = (h->cfg.height + tile_h - 1) / tile_h;
  const uint32_t n = tx * ty;
  PT_HIP(hipStreamSynchronize(h->stream));
  if (h->tiles.cost) PT_HIP(hipFree(h->tiles.cost));
  if (h->d_tile_tmp) PT_HIP(hipFree(h->d_tile_tmp));
  h->tiles = ptd::TileGrid{};
  h->d_tile_tmp = nullptr;
  unsigned long long* cost = nullptr;
  PT_HIP(dev_alloc(&cost, n));
  PT_HIP(dev_alloc(&h->d_tile_tmp, n));
  PT_HIP(hipMemsetAsync(cost, 0, (size_t)n * 8, h->stream));
  h->tiles.tile_w = tile_w; h->tiles.tile_h = tile_h; h->tiles.tiles_x = tx; h->tiles.n_tiles = n; h->tiles.cost = cost;
  return PT_OK;
}

int pt_tile_costs(pt_handle h, uint64_t* host_costs, size_t n_tiles) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  if (!h->tiles.n_tiles) return fail(h, PT_ERR_NOT_READY, "pt_tile_costs_enable has not been called");
  if (!host_costs || n_tiles != h->tiles.n_tiles)
    return fail(h, PT_ERR_INVALID_ARGUMENT, "n_tiles must equal the tile grid's size (" + std::to_string(h->tiles.n_tiles) + ")");
  PT_HIP(hipSetDevice(h->cfg.device));
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "tile costs are 64-bit");
  PT_HIP(hipMemcpyAsync(h->d_tile_tmp, h->tiles.cost, n_tiles * 8, hipMemcpyDeviceToDevice, h->stream));
  if (h->n_items) {
    ptd::TileGrid T = h->tiles;
    T.cost = h->d_tile_tmp;
    hipLaunchKernelGGL(ptd::tile_cost_kernel, dim3((h->n_items + 255) / 256), dim3(256), 0, h->stream, h->n_items, h->acc, T);
    PT_HIP(hipGetLastError());
  }
  PT_HIP(hipMemcpyAsync(host_costs, h->d_tile_tmp, n_tiles * 8, hipMemcpyDeviceToHost, h->stream));
  PT_HIP(hipStreamSynchronize(h->stream));   // host buffer is not touched after return
  return PT_OK;
}

int pt_gather_hdr(pt_handle h, int32_t source, size_t slot_items, float* root_host_bgr) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  // ---- local steps: everything that can fail without a peer happens before this rank joins the exchange
  if (h->comm_broken) return fail(h, PT_ERR_COMM, "the communicator of this handle was aborted: create a new one (pt_comm_init_rank / pt_comm_init_all)");
  if (source != PT_HDR_ACCUMULATORS && source != PT_HDR_FILM) return fail(h, PT_ERR_INVALID_ARGUMENT, "unknown HDR source");
  if (source == PT_HDR_FILM && !h->d_film) return fail(h, PT_ERR_NOT_READY, "no resident film: pt_film_accumulate has not been called");
  if (slot_items < h->n_items || slot_items == 0) return fail(h, PT_ERR_INVALID_ARGUMENT, "slot_items must be >= the rank's work items (and > 0)");
  if (slot_items * 3 >= (1ull << 31)) return fail(h, PT_ERR_INVALID_ARGUMENT, "tile too large");
  PT_HIP(hipSetDevice(h->cfg.device));
  const size_t floats = slot_items * 3;
  if (h->hdr_stage_floats < floats) {
    if (h->d_hdr_stage) PT_HIP(hipFree(h->d_hdr_stage));
    h->d_hdr_stage = nullptr; h->hdr_stage_floats = 0;
    PT_HIP(dev_alloc(&h->d_hdr_stage, floats));
    h->hdr_stage_floats = floats;
  }
  const bool root = h->comm_rank == 0;
  const size_t world = (size_t)h->comm_world;
  const bool exchange = h->comm && world > 1;
  if (root && exchange && h->hdr_gather_floats < world * floats) {
    if (h->d_hdr_gather) PT_HIP(hipFree(h->d_hdr_gather));
    h->d_hdr_gather = nullptr; h->hdr_gather_floats = 0;
    PT_HIP(dev_alloc(&h->d_hdr_gather, world * floats));
    h->hdr_gather_floats = world * floats;
  }
  if (h->n_items < slot_items)
    PT_HIP(hipMemsetAsync(h->d_hdr_stage + 3 * (size_t)h->n_items, 0, (slot_items - h->n_items) * 12, h->stream));
  if (h->n_items) {
    if (source == PT_HDR_FILM) {
      PT_HIP(hipMemcpyAsync(h->d_hdr_stage, h->d_film, (size_t)h->n_items * 12, hipMemcpyDeviceToDevice, h->stream));
    } else {
      hipLaunchKernelGGL(ptd::export_hdr_kernel, dim3((h->n_items + 255) / 256), dim3(256), 0, h->stream, h->n_items, h->acc, h->d_hdr_stage);
      PT_HIP(hipGetLastError());
    }
  }
  const float* result = h->d_hdr_stage;
  if (exchange) {
    const auto deadline = comm_deadline(h);
    // an RCCL call that fails outright leaves the communicator in an unknown state: abort it (the peers time out)
#define PT_NCCL_X(call)                                                                                   \
    do {                                                                                                  \
      const ncclResult_t r_ = (call);                                                                     \
      if (r_ != ncclSuccess && r_ != ncclInProgress) return comm_fail(h, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)
    // ---- the slot size must be the same on every rank (the root's receive counts are its own slot_items): checked
    // once per communicator and slot size with a max all-reduce of {slot, -slot}; every rank sees the same verdict.
    // Device -> host copies are only issued on an IDLE stream (after the polled wait): a copy into pageable host memory
    // blocks the host until the stream reaches it, which must never be behind an exchange a peer may not join.
    if (h->comm_slot_agreed != slot_items) {
      const long long mine[2] = {(long long)slot_items, -(long long)slot_items};
      long long seen[2] = {0, 0};
      PT_HIP(hipMemcpyAsync(h->d_slot_check, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
      PT_HIP(hipStreamSynchronize(h->stream));   // local work only so far; `mine` may go out of scope
      PT_NCCL_X(ncclAllReduce(h->d_slot_check, h->d_slot_check, 2, ncclInt64, ncclMax, h->comm, h->stream));
      if (int rc = comm_wait_host(h, "slot-size agreement", deadline)) return rc;
      if (int rc = comm_wait_stream(h, "slot-size agreement", deadline)) return rc;
      PT_HIP(hipMemcpy(seen, h->d_slot_check, sizeof(seen), hipMemcpyDeviceToHost));
      if (seen[0] != -seen[1])
        return fail(h, PT_ERR_INVALID_ARGUMENT, "slot_items differs between the ranks of the communicator (" + std::to_string(-seen[1]) +
                                                    " .. " + std::to_string(seen[0]) + "); this rank passed " + std::to_string(slot_items));
      h->comm_slot_agreed = slot_items;
    }
    // ---- the gather itself
    if (root) {
      PT_HIP(hipMemcpyAsync(h->d_hdr_gather, h->d_hdr_stage, floats * 4, hipMemcpyDeviceToDevice, h->stream));
      PT_NCCL_X(ncclGroupStart());
      for (size_t r = 1; r < world; ++r) {
        const ncclResult_t e = ncclRecv(h->d_hdr_gather + r * floats, floats, ncclFloat, (int)r, h->comm, h->stream);
        if (e != ncclSuccess && e != ncclInProgress) { (void)ncclGroupEnd(); return comm_fail(h, std::string("ncclRecv: ") + ncclGetErrorString(e)); }
      }
      PT_NCCL_X(ncclGroupEnd());
      result = h->d_hdr_gather;
    } else {
      PT_NCCL_X(ncclSend(h->d_hdr_stage, floats, ncclFloat, 0, h->comm, h->stream));
    }
#undef PT_NCCL_X
    if (int rc = comm_wait_host(h, "HDR gather", deadline)) return rc;     // peers connected, transfer queued on the stream
    if (int rc = comm_wait_stream(h, "HDR gather", deadline)) return rc;   // transfer done (or the communicator aborted)
  } else {
    PT_HIP(hipStreamSynchronize(h->stream));
  }
  if (root && root_host_bgr)   // the stream is idle: this copy cannot wait on anything
    PT_HIP(hipMemcpy(root_host_bgr, result, world * floats * 4, hipMemcpyDeviceToHost));
  return PT_OK;
}

#ifdef PTMI_DIAG_BUILD
// test build only: the NIF launch of batch `batch` of every following pt_path_trace fails (batch < 0: off)
int pt_diag_inject_fault(pt_handle h, int32_t batch) {
  if (!h) return PT_ERR_INVALID_ARGUMENT;
  h->diag_fault_batch = batch;
  return PT_OK;
}

// profiling build only: in-kernel clock of the last stamped fused-NIF launch (nif_kernel_v3 with DIAG bit 5, nif_kernel_v4):
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
