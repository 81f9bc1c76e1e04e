// ptmi_film_comm.h -- resident film, per-tile costs and the RCCL hand-off of HDR tiles (entry points of include/ptmi.h)
// Part of the one translation unit ptmi.hip (host side of include/ptmi.h); included there, in this order:
// ptmi_context.h, ptmi_nif_pack.h, ptmi_nif_launch.h, [the entry points in ptmi.hip], ptmi_film_comm.h.
#pragma once

// ---- multi-GPU film hand-off over RCCL --------------------------------------------------------------------------
// The path shards over pixels with no exchange of ray data (reference: one NIF replica per IPU, "no inter-ipu exchange",
// PathTracerApp.cpp:205-252; results only meet on the host film, AccumulatedImage.cpp:59-74).  The one exchange step is
// this gather of HDR tiles to rank 0 at a save interval: every peer sends its tile straight to the root over its own
// xGMI link (grouped ncclSend / ncclRecv -- never a ring), 12 B per work item.
//
// No call in here can block for ever -- whatever the RCCL underneath does.  Two mechanisms, because two RCCLs are in
// play (pt_runtime_info says which one a process is bound to):
//  (1) Communicators are asked to be NON-BLOCKING (ncclConfig_t::blocking = 0) and their progress is polled with
//      ncclCommGetAsyncError / hipStreamQuery against the handle's deadline (pt_comm_set_timeout, default 120 s).  RCCL
//      2.26.6 (the copy PyTorch ships, what bench.py runs on) honours that: every call returns at once.
//  (2) RCCL 2.27.7 (ROCm 7.2's own, what ipu_trace runs on) does NOT for the calls that need a peer: measured on the GPU
//      box (scripts/diag/rccl_nonblocking.cpp), ncclCommInitRankConfig with blocking = 0 stays inside the call for as long
//      as a rank is missing (its bootstrap sits in a blocking accept()), and ncclCommAbort then blocks too (it joins that
//      initialisation first).  So EVERY RCCL call that may need a peer -- set-up, the exchanges' enqueue, finalize, abort --
//      runs on a helper thread (comm_start) and the caller waits for it against the deadline (comm_join).  A call that
//      does not come back is ABANDONED: its thread is left behind (detached; whatever it was making is released by the
//      thread itself if the call ever returns), the handle loses its communicator and reports PT_ERR_COMM.
// On expiry, on an asynchronous RCCL error, or when another thread asks (pt_comm_abort) the communicator is aborted
// (ncclCommAbort ends the kernels still waiting for a peer), the stream is drained (polled, never trusted), and the call
// returns PT_ERR_COMM; the handle then refuses further gathers until it is given a new communicator.  Every step that can
// fail locally (argument checks, allocations, the export kernel) runs BEFORE a rank enters the exchange, so a rank that
// returns early never leaves its peers inside a collective it has half joined: they time out.

#define PT_NCCL(call)                                                                        \
  do {                                                                                       \
    ncclResult_t r_ = (call);                                                                \
    if (r_ != ncclSuccess && r_ != ncclInProgress) {                                         \
      h->error = std::string(#call) + ": " + ncclGetErrorString(r_);                         \
      return PT_ERR_COMM;                                                                    \
    }                                                                                        \
  } while (0)

using comm_clock = ptw::clock;
using CommJob = ptw::BoundedJob;        // .result holds an ncclResult_t, .made the ncclComm_t's a set-up job is making
using CommWorker = ptw::BoundedWorker;  // (the machinery and why the worker is long-lived: ptmi_comm_worker.h)

static void comm_backoff(unsigned& spins) { ptw::backoff(spins); }
static inline bool nccl_ok(int r) { return r == ncclSuccess || r == ncclInProgress; }

// An RCCL call (or group of calls) on the handle's worker thread; fn(job) returns the ncclResult_t.  A set-up call that comes
// back after it was abandoned has its communicators aborted by the worker itself.
template <class F>
static std::shared_ptr<CommJob> comm_start(pt_handle h, size_t n_comms, F fn) {
  const int device = h->cfg.device;
  return ptw::bounded_start(h->comm_worker, n_comms,
                            [device, fn](CommJob& j) mutable { (void)hipSetDevice(device); return (int)fn(j); },
                            [](CommJob& j) { for (void* c : j.made) if (c) (void)ncclCommAbort(static_cast<ncclComm_t>(c)); });
}

// Wait for a job against a deadline (and pt_comm_abort).  false: abandoned -- the handle has let go of that worker.
static bool comm_join(pt_handle h, CommJob& job, comm_clock::time_point deadline, bool heed_abort_request = true) {
  return ptw::bounded_join(h->comm_worker, job, deadline, heed_abort_request ? &h->comm_abort_req : nullptr);
}

// ncclCommAbort, bounded: it is a blocking call too (see (2) above).  Returns whether the abort has completed; if not it
// goes on in the background on the worker the handle has just let go of.
constexpr int kAbortGraceMs = 3000;
static bool comm_abort_bounded(pt_handle h, ncclComm_t comm) {
  if (!comm) return true;
  auto job = comm_start(h, 0, [comm](CommJob&) { return ncclCommAbort(comm); });
  return comm_join(h, *job, comm_clock::now() + std::chrono::milliseconds(kAbortGraceMs), false);
}

// Abort the handle's communicator and leave the handle without one.  Kernels of this communicator still spinning on a
// peer see the abort flag and exit, so the stream can be drained afterwards.  Returns whether RCCL's abort completed.
static bool comm_abort_now(pt_handle h) {
  const bool finished = comm_abort_bounded(h, h->comm);
  h->comm = nullptr;
  h->comm_broken = true;
  h->comm_slot_agreed = 0;
  h->comm_abort_req.store(false);
  return finished;
}

// Drain the handle's stream after an abort, without trusting it to drain: polled for the grace period.
static bool comm_drain_stream(pt_handle h) {
  const auto until = comm_clock::now() + std::chrono::milliseconds(kAbortGraceMs);
  for (;;) {
    const hipError_t e = hipStreamQuery(h->stream);
    (void)hipGetLastError();
    if (e != hipErrorNotReady) return true;   // idle (or faulted: the next call on the stream reports it)
    if (comm_clock::now() > until) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}

static int comm_fail(pt_handle h, const std::string& why) {
  const bool finished = comm_abort_now(h);
  const bool idle = comm_drain_stream(h);   // nothing of the aborted exchange is left running on the caller's buffers
  h->error = why + " -- communicator aborted";
  if (!finished) h->error += " (RCCL's own abort has not returned after " + std::to_string(kAbortGraceMs) + " ms: left to finish in the background)";
  if (!idle) h->error += " (the handle's stream is still busy with the aborted exchange)";
  return PT_ERR_COMM;
}

static std::string comm_no_progress(pt_handle h, const char* what) {
  if (h->comm_abort_req.load()) return std::string(what) + ": aborted by pt_comm_abort";
  return std::string(what) + ": no progress within " + std::to_string(h->comm_timeout_ms) + " ms (a peer is missing or has failed)";
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
    if (h->comm_abort_req.load() || comm_clock::now() > deadline) return comm_fail(h, comm_no_progress(h, what));
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
      (void)comm_abort_now(h);
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

// Enqueue an exchange on the handle's communicator and stream: `fn` (the RCCL calls; it returns the first failing call's
// result and names it in job.failed_call) runs on a helper thread; then the host side and the device side are awaited.
template <class F>
static int comm_exchange(pt_handle h, const char* what, comm_clock::time_point deadline, F fn) {
  auto job = comm_start(h, 0, fn);
  if (!comm_join(h, *job, deadline)) return comm_fail(h, comm_no_progress(h, what) + "; the RCCL call is still blocked and was left behind");
  if (!nccl_ok(job->result)) return comm_fail(h, std::string(what) + ": " + job->failed_call + ": " + ncclGetErrorString((ncclResult_t)job->result));
  if (int rc = comm_wait_host(h, what, deadline)) return rc;     // peers connected, transfer queued on the stream
  return comm_wait_stream(h, what, deadline);                    // transfer done (or the communicator aborted)
}

// pt_destroy's half of the communicator's life: flush what the communicator still has in flight (ncclCommFinalize, which
// may return ncclInProgress or simply take its time), wait for it and destroy it -- all on a helper thread, bounded by the
// handle's deadline, BEFORE the caller destroys the streams it ran on; a communicator that does not settle is aborted.
static void comm_release(pt_handle h) {
  if (!h->comm) {   // no communicator alive (never made, or aborted to completion): a parked worker has nothing left to keep valid
    if (h->comm_worker) { h->comm_worker->stop(); h->comm_worker.reset(); }
    return;
  }
  const auto deadline = comm_deadline(h);
  ncclComm_t comm = h->comm;
  auto job = comm_start(h, 0, [comm, deadline](CommJob&) {
    ncclResult_t r = ncclCommFinalize(comm);
    unsigned spins = 0;
    while (r == ncclInProgress || r == ncclSuccess) {
      ncclResult_t st = ncclSuccess;
      if (ncclCommGetAsyncError(comm, &st) != ncclSuccess) return ncclInternalError;
      if (st == ncclSuccess) return ncclCommDestroy(comm);
      if (st != ncclInProgress) return st;
      if (comm_clock::now() > deadline) return ncclInProgress;
      comm_backoff(spins);
    }
    return r;
  });
  const bool joined = comm_join(h, *job, deadline + std::chrono::milliseconds(500), false);
  const bool destroyed = joined && job->result == ncclSuccess;
  if (!destroyed) (void)comm_abort_bounded(h, comm);
  else if (h->comm_worker) { h->comm_worker->stop(); h->comm_worker.reset(); }   // nothing of the communicator is left: the worker may go
  h->comm = nullptr;
  h->comm_slot_agreed = 0;
}

extern "C" {

int pt_runtime_info(char* buf, size_t n) {
  if (!buf || n == 0) { g_create_error = "null buffer"; return PT_ERR_INVALID_ARGUMENT; }
  buf[0] = 0;
  auto object_of = [](const void* fn) -> std::string {   // the shared object that DEFINES an imported function
    Dl_info di;
    std::string path = (dladdr(fn, &di) && di.dli_fname) ? di.dli_fname : "";
    if (!path.empty()) { char real[PATH_MAX]; if (realpath(path.c_str(), real)) path = real; }
    std::string out;
    for (char c : path) { if (c == '"' || c == '\\') out += '\\'; out += c; }
    return out;
  };
  int rccl = 0, hip_rt = 0, hip_drv = 0;
  (void)ncclGetVersion(&rccl);
  (void)hipRuntimeGetVersion(&hip_rt);
  (void)hipDriverGetVersion(&hip_drv);
  (void)hipGetLastError();
  const std::string s = "{\"librccl\": \"" + object_of(reinterpret_cast<const void*>(&ncclGetVersion)) + "\", \"libamdhip64\": \"" +
                        object_of(reinterpret_cast<const void*>(&hipRuntimeGetVersion)) + "\", \"rccl_version\": " + std::to_string(rccl) +
                        ", \"rccl_compiled\": " + std::to_string(NCCL_VERSION_CODE) + ", \"hip_runtime_version\": " + std::to_string(hip_rt) +
                        ", \"hip_driver_version\": " + std::to_string(hip_drv) + "}";
  if (s.size() + 1 > n) { g_create_error = "pt_runtime_info: buffer too small"; return PT_ERR_INVALID_ARGUMENT; }
  memcpy(buf, s.c_str(), s.size() + 1);
  return PT_OK;
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

// The RCCL this process bound (librccl.so.1 by SONAME: PyTorch's copy if torch was loaded first, else ROCm's) may be older
// than the headers this file was compiled against.  A non-blocking communicator needs ncclCommInitRankConfig (2.14); an
// older config reader is handed a struct stamped with ITS version, so it never looks for fields it does not know.
static int comm_make_config(ncclConfig_t& cfg, std::string& err) {
  int rt = 0;
  const ncclResult_t r = ncclGetVersion(&rt);
  if (r != ncclSuccess) { err = std::string("ncclGetVersion: ") + ncclGetErrorString(r); return PT_ERR_COMM; }
  if (rt < NCCL_VERSION(2, 14, 0)) {
    err = "the RCCL bound to this process is version " + std::to_string(rt) + ": too old for non-blocking communicators (needs 2.14)";
    return PT_ERR_COMM;
  }
  const ncclConfig_t init = NCCL_CONFIG_INITIALIZER;
  cfg = init;
  cfg.blocking = 0;
  if (rt < NCCL_VERSION_CODE) cfg.version = (unsigned int)rt;
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
  ncclConfig_t cfg;
  if (int rc = comm_make_config(cfg, h->error)) return rc;
  h->comm_broken = false;
  h->comm_abort_req.store(false);
  h->comm_slot_agreed = 0;
  const auto deadline = comm_deadline(h);
  // a rank that never arrives ends here, not in a hang: whether RCCL returns at once (non-blocking honoured: the polled wait
  // below sees the deadline) or stays inside the call until every rank has checked in (then comm_join does)
  auto job = comm_start(h, 1, [id, rank, world, cfg](CommJob& j) mutable {
    return ncclCommInitRankConfig(reinterpret_cast<ncclComm_t*>(&j.made[0]), world, id, rank, &cfg);
  });
  if (!comm_join(h, *job, deadline)) {
    h->comm_broken = true;
    h->comm_abort_req.store(false);
    return fail(h, PT_ERR_COMM, comm_no_progress(h, "communicator set-up") + "; ncclCommInitRankConfig is still blocked and was left behind -- communicator aborted");
  }
  ncclComm_t comm = static_cast<ncclComm_t>(job->made[0]);
  if (!nccl_ok(job->result)) {
    (void)comm_abort_bounded(h, comm);
    return fail(h, PT_ERR_COMM, std::string("ncclCommInitRankConfig: ") + ncclGetErrorString((ncclResult_t)job->result));
  }
  h->comm = comm;
  if (int rc = comm_wait_host(h, "communicator set-up", deadline)) return rc;
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
  ncclConfig_t cfg;
  if (int rc = comm_make_config(cfg, h->error)) return rc;
  const auto deadline = comm_deadline(h);
  // one process, several devices: the rank-wise initialisations form one group (on a helper thread, as every RCCL call
  // that may wait for a peer; here the peers are the other members of the group)
  auto job = comm_start(h, (size_t)n, [devs, n, cfg](CommJob& j) mutable {
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { j.failed_call = "ncclGetUniqueId"; return r; }
    r = ncclGroupStart();
    if (!nccl_ok(r)) { j.failed_call = "ncclGroupStart"; return r; }
    for (int i = 0; i < n; ++i) {
      r = (hipSetDevice(devs[i]) == hipSuccess) ? ncclCommInitRankConfig(reinterpret_cast<ncclComm_t*>(&j.made[i]), n, id, i, &cfg) : ncclUnhandledCudaError;
      if (!nccl_ok(r)) { j.failed_call = "ncclCommInitRankConfig"; (void)ncclGroupEnd(); return r; }
    }
    r = ncclGroupEnd();
    if (!nccl_ok(r)) j.failed_call = "ncclGroupEnd";
    return r;
  });
  if (!comm_join(h, *job, deadline))
    return fail(h, PT_ERR_COMM, comm_no_progress(h, "communicator set-up") + "; the RCCL call is still blocked and was left behind");
  std::vector<ncclComm_t> comms;
  for (void* c : job->made) comms.push_back(static_cast<ncclComm_t>(c));
  auto abort_all = [&]() { for (int i = 0; i < n; ++i) (void)comm_abort_bounded(handles[i], comms[i]); };
  if (!nccl_ok(job->result)) { abort_all(); return fail(h, PT_ERR_COMM, job->failed_call + ": " + ncclGetErrorString((ncclResult_t)job->result)); }
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
  const bool exchange = h->comm != nullptr;   // also at world size 1: the same calls (slot all-reduce, grouped receives -- none --, polled waits) as for N ranks
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
    ncclComm_t comm = h->comm;
    hipStream_t stream = h->stream;
    // ---- the slot size must be the same on every rank (the root's receive counts are its own slot_items): checked
    // once per communicator and slot size with a max all-reduce of {slot, -slot}; every rank sees the same verdict.
    // Device -> host copies are only issued on an IDLE stream (after the polled wait): a copy into pageable host memory
    // blocks the host until the stream reaches it, which must never be behind an exchange a peer may not join.
    if (h->comm_slot_agreed != slot_items) {
      const long long mine[2] = {(long long)slot_items, -(long long)slot_items};
      long long seen[2] = {0, 0};
      PT_HIP(hipMemcpyAsync(h->d_slot_check, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
      PT_HIP(hipStreamSynchronize(h->stream));   // local work only so far; `mine` may go out of scope
      long long* check = h->d_slot_check;
      if (int rc = comm_exchange(h, "slot-size agreement", deadline, [comm, stream, check](CommJob& j) {
            j.failed_call = "ncclAllReduce";
            return ncclAllReduce(check, check, 2, ncclInt64, ncclMax, comm, stream);
          })) return rc;
      PT_HIP(hipMemcpy(seen, h->d_slot_check, sizeof(seen), hipMemcpyDeviceToHost));
      if (seen[0] != -seen[1])
        return fail(h, PT_ERR_INVALID_ARGUMENT, "slot_items differs between the ranks of the communicator (" + std::to_string(-seen[1]) +
                                                    " .. " + std::to_string(seen[0]) + "); this rank passed " + std::to_string(slot_items));
      h->comm_slot_agreed = slot_items;
    }
    // ---- the gather itself (an RCCL call that fails outright leaves the communicator in an unknown state: aborted, the peers time out)
    float* stage = h->d_hdr_stage;
    float* all = h->d_hdr_gather;
    if (root) {
      PT_HIP(hipMemcpyAsync(h->d_hdr_gather, h->d_hdr_stage, floats * 4, hipMemcpyDeviceToDevice, h->stream));
      result = h->d_hdr_gather;
    }
    if (int rc = comm_exchange(h, "HDR gather", deadline, [comm, stream, stage, all, floats, world, root](CommJob& j) {
          if (!root) { j.failed_call = "ncclSend"; return ncclSend(stage, floats, ncclFloat, 0, comm, stream); }
          ncclResult_t r = ncclGroupStart();
          if (!nccl_ok(r)) { j.failed_call = "ncclGroupStart"; return r; }
          for (size_t p = 1; p < world; ++p) {
            r = ncclRecv(all + p * floats, floats, ncclFloat, (int)p, comm, stream);
            if (!nccl_ok(r)) { j.failed_call = "ncclRecv"; (void)ncclGroupEnd(); return r; }
          }
          j.failed_call = "ncclGroupEnd";
          return ncclGroupEnd();
        })) return rc;
  } else {
    PT_HIP(hipStreamSynchronize(h->stream));
  }
  if (root && root_host_bgr)   // the stream is idle: this copy cannot wait on anything
    PT_HIP(hipMemcpy(root_host_bgr, result, world * floats * 4, hipMemcpyDeviceToHost));
  return PT_OK;
}

}  // extern "C"
