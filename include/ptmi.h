/*
 * ptmi.h -- C-ABI of the MI355X path-trace hot path (libptmi.so).
 *
 * Drop-in boundary for markp-gc/ipu_path_trace: each entry point replaces one of the five
 * named Poplar programs / named data streams that PathTracerApp::execute() drives
 * (reference: src/PathTracerApp.cpp:479-483 registers the programs, src/ipu_utils.hpp:288-373
 * StreamableTensor is the stream mechanism).  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary.  All functions return 0 on success and a negative pt_status
 * otherwise; pt_last_error() returns the message (the reference throws std::runtime_error /
 * std::logic_error which GraphManager::run catches once, src/ipu_utils.hpp:532-535 -- the C++
 * shim in ipu_path_trace_amd/host re-throws from these codes).
 *
 * Threading (as the reference, PathTracerApp.cpp:692-709): calls on one handle are made from one
 * thread, strictly setup -> path_trace -> read_results; the library copies from / to the host
 * buffers synchronously and retains no host pointer after returning.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_ABI_VERSION 5

typedef struct pt_context* pt_handle;

/* Wire format of the `trace_buffer` stream: src/codelets/TraceRecord.hpp:7-19
 * (20 bytes; offsets 0/2/4/8/12/16/18).  Padding items carry u = v = 65535
 * (src/LoadBalancer.cpp:66-71).  The reference's tiles trace them like any other item and the film skips them
 * (src/AccumulatedImage.cpp:66); here an item with u >= width or v >= height is NOT traced: its sampleCount advances, its
 * r, g, b and pathLength do not, and pt_stats counts image pixels only. */
typedef struct pt_trace_record {
  uint16_t u, v;
  float r, g, b;
  uint16_t sampleCount;
  uint16_t pathLength;
} pt_trace_record;

enum pt_status {
  PT_OK = 0,
  PT_ERR_INVALID_ARGUMENT = -1,
  PT_ERR_NO_DEVICE = -2,
  PT_ERR_HIP = -3,
  PT_ERR_UNSUPPORTED_MODEL = -4,
  PT_ERR_NOT_READY = -5,
  PT_ERR_OUT_OF_MEMORY = -6,
  PT_ERR_COMM = -7
};

enum { PT_AA_NORMAL = 0, PT_AA_UNIFORM = 1, PT_AA_TRUNCATED_NORMAL = 2 }; /* PathTracerApp.cpp:29-45 */
enum { PT_SAMPLES_HALF = 0, PT_SAMPLES_FLOAT = 1 };                        /* PathTracerApp.cpp:297 */
enum { PT_DTYPE_F16 = 0, PT_DTYPE_F32 = 1 };                                  /* Hdf5Model.cpp:109-133 */

/* Compile-time parameters of the reference graph: the CLI options consumed by
 * PathTracerApp::build() and IpuPathTraceJob::buildGraph() (PathTracerApp.cpp:799-817,
 * IpuPathTraceJob.cpp:95-138). */
typedef struct pt_config {
  uint32_t struct_size;          /* sizeof(pt_config) */
  uint32_t width, height;        /* --width/--height */
  uint32_t max_path_length;      /* --max-path-length (1..64) */
  uint32_t roulette_depth;       /* --roulette-depth (>= 1) */
  float stop_prob;               /* --stop-prob, rounded to half as on the IPU */
  float refractive_index;        /* --refractive-index, rounded to half as on the IPU */
  int32_t aa_noise_type;         /* --aa-noise-type */
  int32_t sample_precision;      /* PT_SAMPLES_HALF reproduces the IPU's half primary samples */
  int32_t device;                /* HIP device ordinal (the role of --ipus device selection) */
  uint32_t max_work_items;       /* capacity of the trace buffer (tiles x rays-per-tile) */
  uint32_t iterations_per_batch; /* sample iterations fused per kernel batch; 0 = auto */
  void* stream;                  /* hipStream_t to run on, or NULL for a private stream */
} pt_config;

/* One dense layer as NifModel streams it (src/neural_networks/DenseLayer.hpp:18-31,
 * NifModel.cpp:375-401): kernel row-major [rows = in][cols = out], bias [cols] or NULL,
 * raw fp16 or fp32 bytes as stored in the Keras H5 (Hdf5Model.cpp:109-133 accepts both; models with float32 layers are
 * slow: the fp32 matrix rate is 1/16 of the fp16 one).
 * Any Dense stack the reference's rule accepts is taken: first layer 4*embedding -> h0, every
 * later layer's input width either the previous width or that + 4*embedding (concat(x, input),
 * NifModel.cpp:305-308), head with 3 outputs; 2..16 layers, embedding 1..16, widths <= 2048. */
typedef struct pt_layer {
  uint32_t rows, cols;
  const void* kernel;
  const void* bias;
  int32_t dtype;                 /* PT_DTYPE_F16 or PT_DTYPE_F32.  A model ALL of whose layers share one type runs as the reference's
                                  * does: the matmul, bias add and ReLU in that type (NifModel.cpp:314-325).  A model that MIXES the two
                                  * is an EXTENSION of this library with cast points of its own: the reference ships no such model, never
                                  * casts x between layers and would most likely refuse one at graph construction.  Here every layer runs
                                  * in the type of its own kernel on the float path (fp32 matrix rate): a binary16 layer rounds its sum
                                  * to half, adds its bias in half and reads its input cast to half (RNE).  No reference fixture exists
                                  * for it: parity unpinned (checked against the oracle's restatement of the same rule only). */
  int32_t relu;                  /* activation == "relu" (NifModel.cpp:323-325) */
} pt_layer;

/* Replaces the three cycle-count streams (PathTracerApp.cpp:598-603) with per-stage device
 * times from HIP events, plus the counters the roofline accounting needs. */
typedef struct pt_stats {
  uint64_t paths;                /* path-samples traced by the last path_trace (work items that are not padding x samples) */
  uint64_t segments;             /* sum of pathLength (LoadBalancer.cpp:198-213 "totalRays") */
  uint64_t escaped;              /* paths that reached the environment light = NIF evaluations */
  uint64_t nif_flops_per_sample; /* NifModel::analyseModel formula (NifModel.cpp:129-133) */
  double path_trace_ms;          /* sum over trace-kernel launches  (path_trace_cycle_count) */
  double nif_ms;                 /* sum over NIF-kernel launches    (nif_cycle_count) */
  double accumulate_ms;          /* sum over accumulate launches */
  double total_ms;               /* whole path_trace program        (iter_cycle_count x iterations) */
  uint32_t trace_launches, nif_launches, accumulate_launches;
  uint32_t first_sample;         /* absolute index of the step's first sample iteration (the RNG is keyed by pixel and this index) */
} pt_stats;

/* One traced path, for kernel-level parity checks: the information the reference keeps in the
 * per-ray contribution stack (PathTracerApp.cpp:301-308) reduced to what the deferred stages use. */
typedef struct pt_path_record {
  uint32_t length;               /* contribution-stack size incl. terminator (codelets.cpp:253) */
  uint32_t escaped;
  float dir[3];                  /* ESCAPED record direction (codelets.cpp:187) */
  float uv[2];                   /* PreProcessEscapedRays output (codelets.cpp:343-347) */
  float throughput[3];           /* product of clr*weight along the path incl. terminal weight */
  float cam[2];                  /* GenerateCameraRays output, half-rounded (codelets.cpp:74-75) */
} pt_path_record;

/* Construct the renderer: the work PathTracerApp::build() + GraphManager compile/load do
 * (PathTracerApp.cpp:310-484, ipu_utils.hpp:473-551).  Fails if no HIP device is usable. */
int pt_create(const pt_config* config, pt_handle* out);
int pt_destroy(pt_handle h);
/* Message for the last failure on `h` (or, with h == NULL, of the last failed pt_create). */
const char* pt_last_error(pt_handle h);
int pt_abi_version(void);
/* ABI 5.  Which shared objects this process really bound the library's HIP and RCCL imports to, and their versions, as one
 * JSON object: {"librccl": path, "libamdhip64": path, "rccl_version": ncclGetVersion() at run time, "rccl_compiled":
 * NCCL_VERSION_CODE of the headers libptmi.so was built against, "hip_runtime_version", "hip_driver_version"}.  libptmi.so
 * imports librccl.so.1 / libamdhip64.so.7 by SONAME: a process that has already loaded other copies under those names
 * (PyTorch ships its own) gets THOSE -- a process holds one HIP runtime, whichever was loaded first.  The reference has no
 * counterpart (its Poplar runtime is one library); the bench line and the tests record this so that nobody certifies one
 * RCCL and measures on another.  Needs no handle and no device.  Always NUL-terminated; PT_ERR_INVALID_ARGUMENT if the
 * buffer is too small (512 bytes are enough). */
int pt_runtime_info(char* buf, size_t n);

/* Program "init_nif_weights" (PathTracerApp.cpp:480; streams NifModel.cpp:375-401): all layer
 * kernels and biases, `max`, and `mean` with -eps already folded in (NifMetaData.cpp:48-53).
 * May be called again to hot-swap the environment (PathTracerApp.cpp:548-557). */
int pt_upload_nif(pt_handle h, const pt_layer* layers, uint32_t n_layers, uint32_t embedding_dim,
                  float max, const float mean[3], int32_t log_tonemap);
/* Constant-radiance environment instead of a NIF (BASELINE config C1; no reference program). */
int pt_set_constant_env(pt_handle h, const float rgb[3]);

/* Program "init_render_settings" (PathTracerApp.cpp:479): streams seed u32[2], anti_alias_scale
 * (half), field_of_view (half, radians), hdri_azimuth (f32, radians), on_device_sample_limit.
 * A new seed restarts the sample-index sequence; the same seed continues it. */
int pt_set_render_settings(pt_handle h, uint64_t seed, float aa_noise_scale, float fov_radians,
                           float azimuth_radians, uint32_t samples_per_step);

/* Program "setup" (PathTracerApp.cpp:481): host -> device copy of the active worklist. */
int pt_setup(pt_handle h, const pt_trace_record* work, size_t n);
/* Program "path_trace" (PathTracerApp.cpp:482): samples_per_step iterations of
 * K2..K12 on the device (PathTracerApp.cpp:432-468).  Blocks until the device is done.
 * On failure every stream of the handle has been drained before the call returns (nothing is
 * left running on buffers the caller may free), the sample sequence has not advanced, and the
 * worklist's accumulators are undefined: call pt_setup again before the next path_trace. */
int pt_path_trace(pt_handle h);
/* Program "read_results" (PathTracerApp.cpp:483): device -> host copy of the worklist plus stats. */
int pt_read_results(pt_handle h, pt_trace_record* work, size_t n, pt_stats* stats);
/* Stats of the last path_trace without the device -> host copy. */
int pt_get_stats(pt_handle h, pt_stats* stats);
/* ABI 4.  Name of the NIF kernel(s) the library dispatched for the uploaded model at its last NIF launch (the reference
 * logs its model through NifModel::analyseModel, NifModel.cpp:122-144; here the kernel choice depends on the shape: fused
 * register-resident, layer by layer, or float32).  Empty before the first launch.  Always NUL-terminated. */
int pt_nif_kernel_name(pt_handle h, char* buf, size_t n);
/* ABI 4.  Calibration of the NIF stage, the counterpart of reading nif_cycle_count (PathTracerApp.cpp:449,598-603) with
 * nothing else on the device: runs the NIF stage of the largest batch of the LAST pt_path_trace again -- same kernel, same
 * compacted queue, which is still resident -- one untimed launch and then `launches` timed ones back to back, with no trace
 * or accumulate kernel beside it.  Returns the average milliseconds per launch (HIP events on the NIF stream) and the
 * number of NIF evaluations per launch (the queue length).  The worklist's accumulators are not touched.  Comparing this
 * rate with the one inside a step separates a slow device from a regression of the pipeline around the kernel. */
int pt_calibrate_nif(pt_handle h, uint32_t launches, double* ms_per_launch, uint64_t* evaluations);

/* Multi-GPU film hand-off.  The path shards over pixels with no exchange of ray data (reference: one NIF
 * replica per IPU, "no inter-ipu exchange of ray data", PathTracerApp.cpp:205-252, shard_utils.cpp:28-38);
 * the only exchange is the film: mean radiance per work item, BGR float32 [n][3] -- the value
 * AccumulatedImage::accumulate adds, AccumulatedImage.cpp:59-74: (b,g,r)/sampleCount, with the device's
 * 32-bit sample count (0 samples -> 0).
 *
 * pt_export_hdr_device writes this rank's tile into a caller-owned DEVICE buffer on the handle's stream.
 *
 * pt_film_accumulate keeps the film on the device between save intervals: it is AccumulatedImage::accumulate
 * (AccumulatedImage.cpp:59-74: film += (b,g,r) * (1/sampleCount)) followed by
 * LoadBalancer::clearInactiveAccumulators (LoadBalancer.cpp:198-213) for every work item, with the host's fp32
 * expressions, so the resident film equals the host film bit for bit.  pt_setup starts a new (zero) film.
 *
 * pt_gather_hdr is the whole hand-off as one call, made by every rank of the communicator: take this rank's
 * tile -- PT_HDR_ACCUMULATORS: mean radiance of the current accumulators (as pt_export_hdr_device);
 * PT_HDR_FILM: the resident film's running sum (the host divides by the step count when it saves,
 * AccumulatedImage.cpp:24,54) -- into a slot of `slot_items` items (>= this rank's item count, equal on all
 * ranks, zero padded), ONE RCCL gather to rank 0 (grouped ncclSend/ncclRecv: every peer sends directly to the
 * root over its own xGMI link), and on rank 0 a copy of all tiles [world][slot_items][3] into `root_host_bgr`
 * (ignored on the other ranks; may be NULL).  Blocking.  Without a communicator it degenerates to export +
 * copy of one tile.
 *
 * Communicators: one rank per handle.  One process per GPU: rank 0 calls pt_comm_get_unique_id, hands the
 * PT_COMM_ID_BYTES bytes to the other processes by any means (MPI, torch.distributed, a file), everyone calls
 * pt_comm_init_rank.  One process driving several GPUs (ipu_trace --ipus N): pt_comm_init_all on the list of
 * handles; pt_gather_hdr is then called from one thread per handle.  pt_destroy releases the communicator.
 *
 * No communicator call blocks for ever (the reference's Poplar engine has no such failure mode: its IPUs are one
 * device, PathTracerApp.cpp:205-252).  Communicators are asked to be non-blocking and their progress is polled against a
 * deadline (pt_comm_set_timeout, default 120000 ms); because the RCCL of ROCm 7.2 (2.27.7) does not honour that for set-up
 * and abort, every RCCL call that may need a peer additionally runs on a worker thread of the handle and is WAITED FOR
 * against the same deadline: a call that never returns is abandoned.  slot_items is checked for
 * agreement between the ranks once per communicator and slot size (PT_ERR_INVALID_ARGUMENT on every rank if it differs).
 * If a peer never joins an exchange (it failed a local check and returned early, crashed, or was never started), if RCCL
 * reports an asynchronous error, or if another thread calls pt_comm_abort(h), the waiting rank aborts its communicator
 * (ncclCommAbort, itself bounded), drains its stream and returns PT_ERR_COMM; the handle then has no communicator and
 * pt_gather_hdr keeps returning PT_ERR_COMM until pt_comm_init_rank / pt_comm_init_all gives it a new one.  CONTRACT for
 * callers that drive several ranks: when one rank's call fails, abort the others (pt_comm_abort -- the only pt_* function
 * that may be called from another thread while a call on the same handle is in progress) or let them time out.
 * pt_comm_init_rank / pt_comm_init_all ask the RCCL they are bound to for its version first (ncclGetVersion): older than
 * 2.14 (no ncclCommInitRankConfig) is refused with PT_ERR_COMM; older than the headers libptmi.so was compiled against,
 * the ncclConfig_t is stamped with the RUNNING library's version, so that library reads exactly the fields it knows
 * (pt_runtime_info and DESIGN.md section 6 say which RCCL a process runs on).
 * Multi-rank exchanges have not been run on hardware yet (no multi-GPU box was available): DESIGN.md section 6. */
enum { PT_HDR_ACCUMULATORS = 0, PT_HDR_FILM = 1 };
#define PT_COMM_ID_BYTES 128
int pt_comm_get_unique_id(void* id_out);
int pt_comm_init_rank(pt_handle h, const void* id, int rank, int world);
int pt_comm_init_all(pt_handle* handles, int n);
int pt_comm_info(pt_handle h, int* rank, int* world);
int pt_comm_set_timeout(pt_handle h, uint32_t milliseconds);
int pt_comm_abort(pt_handle h);
int pt_film_accumulate(pt_handle h);
/* Path-length balancing without the worklist leaving the device.  The reference re-deals work by the pathLength every
 * step returns per work item (LoadBalancer::allocateWorkByPathLength, LoadBalancer.cpp:141-192), which costs the whole
 * trace buffer both ways per step; across GPUs the unit of re-dealing is an image tile, so what the balancer needs is
 * kilobytes: per tile of tile_w x tile_h pixels (row-major grid over width x height) the sum of pathLength of this
 * handle's work items.  pt_tile_costs_enable starts the bookkeeping (what pt_film_accumulate clears from the
 * accumulators is folded into the per-tile sums first); pt_tile_costs copies, for all n_tiles = ceil(width / tile_w) x
 * ceil(height / tile_h) tiles, tracked sums + the current accumulators' pathLength to the host (uint64 each; padding
 * items, u = v = 65535, belong to no tile).  pt_setup starts the sums afresh. */
/* pt_film_seed: after a re-deal the film has to follow its pixels.  Sets the resident film of the n = current work
 * items from host values (BGR float32 [n][3], the running sums pt_gather_hdr(PT_HDR_FILM) returned for those pixels),
 * so that every pixel's fp32 sum continues in step order whichever handle owns it: a balanced render equals the
 * unbalanced one bit for bit.  Called after pt_setup (which zeroes the film), at save intervals only. */
int pt_film_seed(pt_handle h, const float* host_bgr, size_t n);
int pt_tile_costs_enable(pt_handle h, uint32_t tile_w, uint32_t tile_h);
int pt_tile_costs(pt_handle h, uint64_t* host_costs, size_t n_tiles);
int pt_gather_hdr(pt_handle h, int32_t source, size_t slot_items, float* root_host_bgr);
int pt_export_hdr_device(pt_handle h, void* device_bgr, size_t n);
/* Clear r,g,b,sampleCount,pathLength on the device worklist
 * (LoadBalancer::clearInactiveAccumulators, LoadBalancer.cpp:198-213) without a host round trip. */
int pt_clear_accumulators(pt_handle h);
int pt_synchronize(pt_handle h);

/* Standalone NIF inference, host buffers: u, v in [0,1) -> decoded BGR float32 [n][3]
 * (NifModel's streamed-IO mode, NifModel.cpp:268-278,338-350). */
int pt_nif_infer(pt_handle h, const float* u, const float* v, size_t n, float* bgr);
/* Trace individual paths (pixel u,v; absolute sample index) and return their records. */
int pt_trace_paths(pt_handle h, const uint16_t* u, const uint16_t* v, const uint32_t* sample_index,
                   size_t n, pt_path_record* out);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
