// ptmi_context.h -- per-handle device state (pt_context), error helpers, trace-grid geometry
// Part of the one translation unit ptmi.hip (host side of include/ptmi.h); included there, in this order:
// ptmi_context.h, ptmi_nif_pack.h, ptmi_nif_launch.h, [the entry points in ptmi.hip], ptmi_film_comm.h.
#pragma once

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
  uint32_t n_real = 0;                       // work items that are not padding (u < width and v < height: AccumulatedImage.cpp:66)
  unsigned long long* d_counters = nullptr;  // [0] segments, [1] escaped
  unsigned long long* h_counters = nullptr;  // the same in pinned host memory: copied on the stream at the end of a step (no blocking hipMemcpy)

  // batch buffers, double-buffered: the trace kernel of batch b+1 runs on `trace_stream` while the NIF
  // kernel of batch b (MFMA-bound) runs on `stream`
  uint32_t iters_per_batch = 1;
  uint32_t first_batch_iters = 1;   // iterations of a step's first batch (see enqueue_path_trace)
  size_t batch_paths_cap = 0;
  size_t queue_cap = 0;
  // Grid of the persistent trace kernel: six workgroups per CU (1536 on an MI355X).  Measured, scripts/sweep_trace_blocks.py:
  // 9.1 ms per 331 M-path step for 1280...1536 workgroups against 10.2 for 1600...2048 (the grid of rounds 1-3), the C2 step
  // unchanged -- the 60-VGPR / 106-SGPR kernel is resident six-fold per CU, not eight-fold as
  // hipOccupancyMaxActiveBlocksPerMultiprocessor reports, so a larger grid runs its last workgroups on part of the chip.
  static constexpr int kTraceBlocksPerCu = 6;
  // Bytes of batch buffers per path of batch capacity, both sets: per path plen 1 + rad 12; per queue slot (one per path,
  // rounded up per region) q_* 24 + survivor note 16 + path state 48.
  static constexpr uint64_t kBatchBytesPerPath = 2 * (1 + 12 + 24 + 16 + 48);
  uint32_t trace_blocks = 1536;
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
    // geometry of the last batch whose NIF stage ran on this set (pt_calibrate_nif replays the larger one): 0 = none yet
    uint32_t last_paths = 0, last_regions = 0, last_region_cap = 0;
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
  // float32 models (pt_nif_f32.h): padded row-major kernels and biases of all layers in one buffer, chunk buffers
  bool nif_f32 = false;
  struct F32Layer { size_t w_off, b_off; uint32_t k_act, k_in, ldw, relu, half_out, cast_half; };
  std::vector<F32Layer> f32_layers;
  float* d_f32_weights = nullptr;
  float* d_f32_act[2] = {nullptr, nullptr};
  float* d_f32_feat = nullptr;
  uint32_t f32_chunk = 0, f32_lda = 0, f32_ldf = 0;
  float4* d_head_partial = nullptr;   // fused head: [2 FB][chunk samples] partial sums
  float4* d_head_in = nullptr;        // head weights of the Fourier-feature inputs [4][E], if the head concatenates them
  float head_bias[3] = {0, 0, 0};
  uint32_t head_piece_base = 0;
  ptd::NifParams nif{};
  uint4* d_wpack = nullptr;
  uint4* d_bpack = nullptr;
  uint64_t nif_flops = 0;
  std::string nif_kernel;   // what launch_nif dispatched last (pt_nif_kernel_name): the bench line quotes the library, not a guess
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

  // stats.  The per-stage times are read lazily (pt_get_stats / pt_read_results): 3 hipEventElapsedTime calls per batch are
  // host time a step of a small image should not pay (BASELINE configs[0] is one millisecond of device work per step).
  pt_stats stats{};
  std::vector<hipEvent_t> events;
  struct StageSpan { size_t a, b; int kind; };   // event pair around one stage of one batch: 0 trace, 1 NIF, 2 accumulate
  std::vector<StageSpan> spans;
  size_t e_begin_i = 0, e_end_i = 0;
  bool spans_pending = false;

  // scratch for the standalone entry points
  void* d_scratch = nullptr;
  size_t scratch_bytes = 0;

  // multi-GPU film hand-off: RCCL communicator (one rank per handle) and the HDR tile buffers
  ncclComm_t comm = nullptr;
  std::shared_ptr<ptw::BoundedWorker> comm_worker;   // the long-lived thread that makes the handle's RCCL calls (ptmi_comm_worker.h): created with the first communicator call, dropped when a call never returns
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

inline int hip_status(hipError_t e);

#define PT_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      h->error = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return hip_status(e_);                                                                 \
    }                                                                                        \
  } while (0)

// A failed allocation is its own status (include/ptmi.h): the caller can retry with a smaller max_work_items.
inline int hip_status(hipError_t e) {
  if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return PT_ERR_OUT_OF_MEMORY; }   // not sticky: clear it for the next call
  return PT_ERR_HIP;
}

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
  struct Src { int disc; float c[3]; float r; float col[3]; int type; };
  Src src[ptd::kNumObjects];
  for (int i = 0; i < ptd::kNumObjects; ++i) {   // ONE table: the kernels' compile-time scene (pt_trace.h::scene_const, codelets.cpp:111-144)
    const ptd::SceneConst S = ptd::scene_const(i);
    src[i] = Src{S.disc ? 1 : 0, {S.cx, S.cy, S.cz}, S.radius, {S.colr, S.colg, S.colb}, S.type};
  }
  for (int i = 0; i < ptd::kNumObjects; ++i) {
    ptd::SceneObject& o = P.obj[i];
    o.cx = src[i].c[0]; o.cy = src[i].c[1]; o.cz = src[i].c[2];
    o.radius = src[i].r;
    o.r2 = src[i].r * src[i].r;
    o.nx = 0.f; o.ny = src[i].disc ? 1.f : 0.f; o.nz = 0.f;
    o.colr = src[i].col[0]; o.colg = src[i].col[1]; o.colb = src[i].col[2];
    o.type = src[i].type;
    o.is_disc = src[i].disc;
    // constants of a ray that starts at the origin, by the device's own expressions (this file is compiled with
    // -ffp-contract=off like the kernels; volatile keeps every intermediate a rounded binary32 whatever the host's FLT_EVAL_METHOD)
    volatile float ox = 0.f - o.cx, oy = 0.f - o.cy, oz = 0.f - o.cz;          // sub(o, c)
    volatile float xx = ox * ox, yy = oy * oy, zz = oz * oz;
    volatile float d0 = xx + yy, d1 = d0 + zz;                                 // dot(oc, oc), left to right
    volatile float cc = d1 - o.r2;
    o.ocx = ox; o.ocy = oy; o.ocz = oz;
    o.c4 = 4.0f * cc;
    volatile float kx = (o.cx - 0.f) * o.nx, ky = (o.cy - 0.f) * o.ny, kz = (o.cz - 0.f) * o.nz;   // dot(sub(c, o), n)
    volatile float k0 = kx + ky, k1 = k0 + kz;
    o.kdisc = k1;
    o.same_centre = (i > 0 && !src[i].disc && !src[i - 1].disc && src[i].c[0] == src[i - 1].c[0] && src[i].c[1] == src[i - 1].c[1] &&
                     src[i].c[2] == src[i - 1].c[2]) ? 1 : 0;
  }
}

// Round-up reciprocal of the work-item count: (x * magic) >> shift == x / n for every x < 2^31 (Granlund & Montgomery: with
// s = ceil(log2 n) and magic = floor(2^(31+s) / n) + 1 the error magic * n - 2^(31+s) lies in (0, 2^s], so the product's excess
// over x / n stays below 1 / n).  The batch size keeps path indices below 2^31 (pt_create).
void item_divider(uint32_t n, uint32_t& magic, uint32_t& shift) {
  uint32_t s = 0;
  while ((1ull << s) < n) ++s;
  const uint64_t m = ((1ull << (31 + s)) / n) + 1ull;
  magic = (uint32_t)m;     // < 2^32: n > 2^(s-1)
  shift = 31 + s;
}

void fill_trace_params(pt_handle h, ptd::TraceParams& P) {
  memset(&P, 0, sizeof(P));
  fill_scene(P);
  const pt_config& c = h->cfg;
  const float w = (float)c.width, hgt = (float)c.height;
  const float fov = host_hround(h->fov);        // field_of_view stream is half (PathTracerApp.cpp:591)
  P.width_f = w;
  P.height_f = hgt;
  P.width = c.width;
  P.height = c.height;
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
  item_divider(h->n_items ? h->n_items : 1u, P.div_magic, P.div_shift);
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
// `cap` = pt_context::trace_blocks: the workgroups of the persistent trace kernel that are resident at once (pt_create).
TraceGrid trace_grid(uint32_t total, uint32_t cap) {
  const uint32_t n_chunks = (total + 63u) / 64u;
  uint32_t blocks = (n_chunks + 3u) / 4u;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  TraceGrid g;
  g.blocks = blocks;
  g.n_waves = blocks * 4u;
  g.region_cap = 4u * ((n_chunks + g.n_waves - 1u) / g.n_waves) * 64u;
  return g;
}

}  // namespace
