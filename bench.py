#!/usr/bin/env python3
"""Headline benchmark: Mpath-samples/sec @1104x1000, 300 spp/step, depth 8 (BASELINE.json).

One "step" = one `path_trace` program (reference: src/PathTracerApp.cpp:693) over the resident
worklist: samples-per-step iterations of ray-gen -> path-trace -> NIF -> accumulate.  The metric is
the reference's own: W*H*spp_step / seconds (src/PathTracerApp.cpp:766-767), in millions.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  The image is tile-partitioned
(ipu_path_trace_amd/partition.py), every rank traces its tiles with no data-path collective, and the
HDR tiles are gathered to rank 0 once per save interval (here: once, after the last timed step).
Total work is fixed as N grows -> "scaling": "strong".

`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) starts its own N ranks: the parent, before
it touches the GPU in any way, runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ... bench.py <same arguments>` as a CHILD process, relays its output (rank 0's JSON line) and exits with its
code -- the reference's `--ipus N` is one command too (src/main.cpp:17-19, src/PathTracerApp.cpp:205-252).

`--dist` (or BENCH_FORCE_DIST=1) makes `--gpus 1` take every branch of an N-rank run with the REAL backend: a child started
by spawn_ranks(1), init_process_group("nccl"), the ncclUniqueId from rank 0 over torch.distributed, pt_comm_init_rank(id, 0,
1), the all-reduced agreement, the warm-up product gather with its fallback switch, max-over-ranks on a CUDA tensor -- all of
C4's orchestration that one GPU can execute (tests/test_multi_gpu_launch.py).  The bench line's `runtime` object says which
librccl / libamdhip64 the process really bound (pt_runtime_info): torch is imported first here, so libptmi.so's imports of
librccl.so.1 / libamdhip64.so.7 resolve to the copies PyTorch ships -- ONE HIP runtime per process, at every N.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16


HBM_PEAK_GBPS = 8000.0                        # same guide: HBM3E ~8 TB/s


def _profiles(pattern):
    """Committed profile summaries matching `pattern`, newest round first."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            with open(path) as f:
                yield json.load(f), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue


def hbm_traffic_from_profile(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC passes (counters cannot be read inside this process).
    Returns (bytes, file) -- the file is named in the bench line so nobody takes the figure for a measurement of this run."""
    for doc, path in _profiles("r*_pmc.json"):
        for name, entry in doc.items():
            if isinstance(entry, dict) and name.startswith(kernel_prefix) and "hbm_bytes_per_launch_corrected" in entry:
                return entry["hbm_bytes_per_launch_corrected"], path
    return None, None


def trace_counters_from_profile(depth=8):
    """Wave-level VALU instructions per path of trace_kernel, from the committed PMC pass of scripts/pmc_trace.sh taken at
    this --max-path-length (files without the field are depth 8)."""
    for doc, path in _profiles("r*_trace_pmc.json"):
        if "valu_wave_instr_per_path" in doc and doc.get("max_path_length", 8) == depth:
            return doc, path
    return None, None


def secondary_configs(ptmi, nif_assets, W, H, depth, meta, mean, spp=300, steps=3):
    """Two more configurations measured in this run on the same image -- never the bench `value`, reported so that the
    numbers quoted for them in DESIGN.md / README.md can be checked against a driver-run line: BASELINE config C5 (NIF
    8 x 1024 fp16, the layer-by-layer path, 14,891,011 FLOP per evaluation) and the same 6 x 320 network with float32
    variables (the float path, pt_nif_f32.h).  One warm-up step (30 spp), then `steps` timed steps of `spp` samples per pixel
    -- the 300 spp per step BASELINE states -- each; `value` is the MEDIAN step, min / max beside it."""
    out = {}
    for name, kw, peak in (("c5_nif_8x1024_fp16", dict(hidden=1024, layer_count=8), MFMA_F16_DENSE_PEAK_TFLOPS),
                           ("nif_6x320_float32", dict(hidden=320, layer_count=6, dtype=np.float32), 157.3)):
        try:
            L = nif_assets.synthetic_nif(embedding_dim=meta["embedding_dimension"], **kw)
            r = ptmi.Renderer(W, H, max_path_length=depth)
            r.init_nif_weights(L, meta["embedding_dimension"], meta["max"], mean)
            r.init_render_settings(samples_per_step=min(spp, 30))
            r.setup(ptmi.worklist(W, H))
            r.path_trace()
            r.init_render_settings(samples_per_step=spp)
            rates, tfs, mss = [], [], []
            for _ in range(steps):
                t = time.perf_counter()
                r.path_trace()
                dt = time.perf_counter() - t
                st = r.stats()
                rates.append(st.paths / dt / 1e6)
                mss.append(dt * 1e3)
                tfs.append(st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12)
            kname = r.nif_kernel_name()
            r.close()
            tf = float(np.median(tfs))
            out[name] = {"value": float(np.median(rates)), "min": min(rates), "max": max(rates), "unit": "Mpath-samples/s",
                         "steps": steps, "spp_per_step": spp, "ms_per_step": float(np.median(mss)),
                         "nif_flops_per_sample": int(st.nif_flops_per_sample), "nif_tflops": tf, "nif_tflops_min": min(tfs),
                         "peak_tflops": peak, "frac": tf / peak, "kernel": kname}
        except Exception as e:   # noqa: BLE001 -- a secondary figure must never cost the headline line
            out[name] = {"error": str(e)}
    return out


def c1_on_the_gpu(ptmi, steps=200):
    """BASELINE configs[0] (256x256, 16 spp, depth 4, constant sky) on the GPU, for the table beside the CPU's rate.  One step
    is 1 M paths -- about 0.1 ms of device work -- so what is measured is the host's cost per pt_path_trace call.  Timed
    BEFORE any OpenMP region of this process exists (the oracle's 100+ worker threads keep spinning for a while after a
    parallel region and would be timed instead): round 4's driver-run line had this leg 9x slower than the builder's for that
    reason.  Host wall clock and the device's own time (pt_stats.total_ms: HIP events around the step) are both reported."""
    g = ptmi.Renderer(256, 256, max_path_length=4)
    g.set_constant_env((1.0, 1.0, 1.0))
    g.init_render_settings(samples_per_step=16)
    g.setup(ptmi.worklist(256, 256))
    for _ in range(10):
        g.path_trace()
    t = time.perf_counter()
    for _ in range(steps):
        g.path_trace()
    gdt = time.perf_counter() - t
    dev_ms = []
    for _ in range(20):                      # reading the stage times costs host time: outside the timed loop
        g.path_trace()
        dev_ms.append(g.stats().total_ms)
    g.close()
    paths = 256 * 256 * 16
    return {"gpu_value": steps * paths / gdt / 1e6, "gpu_ms_per_step": gdt / steps * 1e3, "gpu_steps": steps,
            "gpu_device_ms_per_step": float(np.median(dev_ms)), "gpu_value_device_time": paths / (float(np.median(dev_ms)) * 1e-3) / 1e6,
            "gpu_what": "host wall over %d back-to-back pt_path_trace calls, measured before the CPU legs start their OpenMP "
                        "threads; device time = pt_stats.total_ms (HIP events), median of 20 further steps" % steps}


def cpu_baseline(width, height, depth, layers, meta, mean, c1_gpu, target_seconds=12.0):
    """The CPU beside the GPU number: the oracle's source -- a restatement, upstream
    external/light is not vendored -- in its TIMING build (oracle/Makefile: -O3 -march=native -fopenmp, contraction allowed,
    F16C conversions, NIF batches of 64 through a register-blocked AVX2 / AVX-512 kernel), compiled on this host, on all of
    its cores.  Two legs: `c1` = BASELINE configs[0] in full (256x256, 16 spp, depth 4, constant sky; repeated to fill a few
    seconds; the GPU's rate on the same config beside it) and `c2_shape` = a bounded pixel sample of the benchmark's own
    workload (same image, depth and NIF).  The strict build (-ffp-contract=off, the parity checker) is timed on a small
    sample for reference.  `value` is the c2_shape rate: the same workload as the bench `value`."""
    from oracle import pt_oracle as O
    O.build()
    use_fast, fast_error = True, None
    try:
        fast = O.lib(fast=True)                          # builds libpt_oracle_fast.so for THIS host if need be
    except Exception as e:   # noqa: BLE001 -- a host without AVX2 / F16C / FMA (oracle/Makefile's #error) or any other compile
        # failure: the strict build is timed instead and the line says so -- a secondary figure never costs the headline
        use_fast, fast_error, fast = False, str(e), O.lib()
    cores = int(fast.orc_max_threads())
    info = fast.orc_build_info().decode()

    # ---- c1: 256 x 256, 16 spp, depth 4, constant sky, seed 1 (no NIF)
    c1_cfg = O.make_config(width=256, height=256, max_path_length=4, env_mode=O.ENV_CONSTANT, env_rgb=(1.0, 1.0, 1.0))
    work = O.worklist(256, 256)
    O.render(c1_cfg, None, work, 0, 16, fast=use_fast)      # warm-up (threads, page faults)
    reps, paths, t = 0, 0, time.perf_counter()
    while True:
        st = O.render(c1_cfg, None, work, 16 * (reps + 1), 16, fast=use_fast)
        reps += 1
        paths += st.paths
        dt = time.perf_counter() - t
        if dt > 3.0 or reps >= 2000:
            break
    c1 = {"value": paths / dt / 1e6, "unit": "Mpath-samples/s", "seconds": dt,
          "sample": "configs[0] in full: 256x256, 16 spp, depth 4, constant sky, %d repetitions" % reps}
    c1.update(c1_gpu)                                    # the GPU on the same config, measured earlier (c1_on_the_gpu)

    # ---- c2_shape: bounded pixel sample of the benchmark's workload
    nif = O.Nif(layers, meta["embedding_dimension"], meta["max"], mean, fast=use_fast)
    cfg = O.make_config(width=width, height=height, max_path_length=depth, env_mode=O.ENV_NIF)
    full = O.worklist(width, height)
    rng = np.random.default_rng(0)
    probe = full[rng.choice(full.size, min(full.size, 200000), replace=False)].copy()
    t = time.perf_counter()
    O.render(cfg, nif, probe, 0, 1, fast=use_fast)
    rate = probe.size / (time.perf_counter() - t)
    n = int(min(full.size, max(20000, rate * target_seconds)))
    sample = full[rng.choice(full.size, n, replace=False)].copy()
    spp = int(min(64, max(1, round(rate * target_seconds / n))))
    t = time.perf_counter()
    st = O.render(cfg, nif, sample, 0, spp, fast=use_fast)
    dt = time.perf_counter() - t
    flops = nif.flops_per_sample()
    c2 = {"value": st.paths / dt / 1e6, "unit": "Mpath-samples/s", "seconds": dt,
          "nif_gflops": st.escaped * flops / dt / 1e9,
          "sample": "%d random pixels of the %dx%d image x %d spp, depth %d, same synthetic NIF" % (n, width, height, spp, depth)}

    # ---- the strict build (the parity checker) on a small sample, for reference
    snif = O.Nif(layers, meta["embedding_dimension"], meta["max"], mean)
    small = full[rng.choice(full.size, min(full.size, max(20000, 160 * cores)), replace=False)].copy()
    t = time.perf_counter()
    sst = O.render(cfg, snif, small, 0, 1)
    sdt = time.perf_counter() - t
    return {"value": c2["value"], "unit": "Mpath-samples/s", "cores": cores, "kind": "port",
            "sample": c2["sample"] + " (%.1f s of CPU work)" % dt,
            "build_flags": info, "nproc": os.cpu_count(), "omp_max_threads": cores, "timing_build_error": fast_error,
            "c1": c1, "c2_shape": c2,
            "strict_build": {"value": sst.paths / sdt / 1e6, "unit": "Mpath-samples/s", "seconds": sdt,
                             "build_flags": O.lib().orc_build_info().decode(),
                             "sample": "%d random pixels x 1 spp of the c2_shape workload" % small.size},
            "what": "the oracle's source (a restatement of the reference's path; upstream external/light is not vendored) in its "
                    "timing build on this host's cores; never the parity checker"}


def spawn_ranks(n):
    """Launch this script as n ranks under torch.distributed.run, as a child process; relay stdout and the exit code.
    Nothing in this (parent) process has initialised the GPU: no exec of a GPU process, only a child."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1104)
    ap.add_argument("--height", type=int, default=1000)
    ap.add_argument("--samples-per-step", type=int, default=300)
    ap.add_argument("--max-path-length", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=320)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the two secondary measurements (BASELINE config C5, float32 NIF) reported beside the headline")
    # the reference's options of the same names (PathTracerApp.cpp:794-830); defaults keep the headline run unchanged
    ap.add_argument("--save-interval", type=int, default=0,
                    help="gather the HDR tiles to rank 0 every N steps (0: once, after the last timed step)")
    ap.add_argument("--enable-load-balancing", action="store_true",
                    help="re-deal image tiles between ranks by measured path length at every save interval")
    ap.add_argument("--dump-film", default="",
                    help="rank 0 writes the final film (float32 H x W x 3 BGR, mean radiance) to this .npy file; at one GPU the "
                         "film takes the same hand-off path (pt_gather_hdr of one tile) -- used by the tests to compare N ranks with one")
    ap.add_argument("--dist", action="store_true",
                    help="take the N-rank code path whatever --gpus says (with --gpus 1: everything an N-rank run does -- launcher, "
                         "nccl process group, communicator from a broadcast unique id, product gather, max over ranks -- on one "
                         "GPU); BENCH_FORCE_DIST=1 does the same")
    ap.add_argument("--comm-fault", default="", choices=["", "corrupt-id"],
                    help="TEST ONLY: damage the ncclUniqueId before pt_comm_init_rank, to exercise the agreed fallback to "
                         "torch.distributed's gather (the bench line then says so)")
    ap.add_argument("--comm-timeout-ms", type=int, default=60000,
                    help="deadline of the communicator set-up and of every gather (pt_comm_set_timeout)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    dist_mode = args.dist or os.environ.get("BENCH_FORCE_DIST") == "1"
    if (args.gpus > 1 or dist_mode) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BENCH_REHEARSAL=1: every rank shares GPU 0 and the collective runs over gloo -- only for exercising the N > 1
    # code path on a one-GPU box; never a measurement.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    multi = world > 1 or dist_mode            # the N-rank code path (at world size 1 only with --dist)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from ipu_path_trace_amd import nif_assets, partition, ptmi

    W, H, spp, depth = args.width, args.height, args.samples_per_step, args.max_path_length
    meta = dict(nif_assets.URBAN_ALLEY_META)
    mean = nif_assets.folded_mean(meta)
    layers = nif_assets.synthetic_nif(hidden=args.hidden, layer_count=args.layers,
                                      embedding_dim=meta["embedding_dimension"])

    owner = partition.round_robin_owner(W, H, world)
    work = partition.worklist_for_owner(W, H, owner, rank)
    redeal = args.enable_load_balancing and multi and args.save_interval > 0
    counts = [partition.max_items_per_rank(W, H, world)] if redeal else partition.items_per_rank(W, H, world)
    stream = torch.cuda.current_stream().cuda_stream
    r = ptmi.Renderer(W, H, max_work_items=max(counts), max_path_length=depth, device=device_index, stream=stream)
    r.init_nif_weights(layers, meta["embedding_dimension"], meta["max"], mean)   # program init_nif_weights
    r.init_render_settings(seed=1, aa_noise_scale=0.3, fov_degrees=90.0, samples_per_step=spp)
    if redeal:
        r.tile_costs_enable(partition.TILE, partition.TILE)
    r.setup(work)                                                                 # inputs resident in HBM
    slot = max(counts)
    hdr = torch.empty((slot, 3), dtype=torch.float32, device="cuda")
    gathered = {"tiles": None, "via": "none (one GPU)"}

    # The HDR-tile gather is part of the product: libptmi.so owns an RCCL communicator (one rank per handle) and
    # pt_gather_hdr sends every rank's tile straight to rank 0.  The ncclUniqueId travels over torch.distributed, which
    # the driver's launcher has set up anyway.  Should the communicator fail to come up on some rank, every rank falls
    # back to torch.distributed's gather of the same device buffers and the bench line says so.
    product_gather = False
    COMM_TIMEOUT_MS = args.comm_timeout_ms   # a rank whose peers never arrive gives up after this long and everyone takes the fallback

    def everyone(ok):
        flag = torch.tensor([int(ok)], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    if multi and not rehearsal:
        ok, why, uid = 1, "", None
        if rank == 0:
            try:
                uid = ptmi.comm_unique_id()
                if args.comm_fault == "corrupt-id":
                    # TEST ONLY: an id no bootstrap root listens behind (ncclBootstrapHandle = magic u64, then the root's
                    # socket address: the two port bytes are flipped)
                    uid = uid[:10] + bytes([uid[10] ^ 0x5A, uid[11] ^ 0xA5]) + uid[12:]
            except Exception as e:   # noqa: BLE001 -- the other ranks still have to be told (they wait in the broadcast)
                why = str(e)
        ids = [uid]
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is None:
            ok = 0
        else:
            try:
                r.comm_set_timeout(COMM_TIMEOUT_MS)
                r.comm_init_rank(ids[0], rank, world)   # non-blocking set-up polled against the deadline: never a hang
            except Exception as e:   # noqa: BLE001 -- any failure means "use the fallback", on every rank
                ok, why = 0, str(e)
        product_gather = everyone(ok)
        gathered["via"] = ("pt_gather_hdr (RCCL inside libptmi.so)" if product_gather
                           else "torch.distributed gather (pt_comm_init_rank failed: %s)" % (why or "on another rank"))
    elif multi:
        gathered["via"] = "gloo via host memory (rehearsal)"

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_hdr(n_items):
        """One gather of HDR tiles to rank 0: mean BGR of the current accumulators, [world][slot][3] on rank 0."""
        if not multi:
            if args.dump_film:
                gathered["tiles"] = r.gather_hdr(slot)                  # no communicator: export + copy of the one tile
            return
        if product_gather:
            gathered["tiles"] = r.gather_hdr(slot)                      # export + RCCL gather + copy to the host
            return
        fallback_gather(n_items)

    def fallback_gather(n_items):
        r.export_hdr_device(hdr.data_ptr(), n_items)
        if rehearsal:
            torch.cuda.synchronize()
            host = hdr.cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
        else:
            parts = [torch.empty_like(hdr) for _ in range(world)] if rank == 0 else None
            dist.gather(hdr, parts, dst=0)
        if rank == 0:
            gathered["tiles"] = np.stack([p_.cpu().numpy() for p_ in parts])

    film_sum = np.zeros((H, W, 3), dtype=np.float64) if rank == 0 else None   # sum over intervals of mean x steps
    if args.dump_film and not multi:
        gathered["via"] = "pt_gather_hdr of one tile (no communicator)"
    state = {"owner": owner, "work": work, "steps_in_interval": 0}

    def hand_off(last):
        """Save-interval film hand-off (AccumulatedImage::accumulate, AccumulatedImage.cpp:59-74): mean BGR per work
        item -> one gather of HDR tiles to rank 0; optionally re-deal tiles by path length (N3) and start afresh."""
        gather_hdr(state["work"].size)
        if rank == 0 and (multi or args.dump_film) and (args.save_interval > 0):
            film = partition.assemble_hdr(W, H, world, gathered["tiles"], owner=state["owner"])
            film_sum[...] += film.astype(np.float64) * state["steps_in_interval"]
        if last or args.save_interval <= 0:
            return
        if redeal:
            # per-tile path-length sums of the interval, summed on the device: 8 B per 16x16 tile leave it, not the worklist
            cost = torch.from_numpy(r.tile_costs(W, H).astype(np.float64))
            if not rehearsal:
                cost = cost.cuda()
            dist.all_reduce(cost, op=dist.ReduceOp.SUM)
            state["owner"] = partition.deal_by_path_length(cost.cpu().numpy(), world)
        state["work"] = partition.worklist_for_owner(W, H, state["owner"], rank)
        r.setup(state["work"])                                                # zeroed accumulators for the next interval
        state["steps_in_interval"] = 0

    for _ in range(args.warmup):
        r.path_trace()
    if multi and (args.warmup or product_gather):
        # untimed: the first gather sets up RCCL's point-to-point channels (lazily, on first use); the timed hand-off
        # must measure the transfer, not the connection set-up.  It is also the proof that the product gather works on
        # this node: if it fails or times out on ANY rank, every rank switches to torch.distributed's gather.
        if product_gather:
            ok, why = 1, ""
            try:
                gather_hdr(work.size)
            except ptmi.PtError as e:
                ok, why = 0, str(e)
            if not everyone(ok):
                product_gather = False
                gathered["via"] = "torch.distributed gather (pt_gather_hdr failed in the warm-up: %s)" % (why or "on another rank")
                fallback_gather(work.size)
        else:
            gather_hdr(work.size)
    if args.warmup and args.save_interval > 0:
        r.setup(work)                                                         # intervals count timed steps only
    agg = {"escaped": 0, "segments": 0, "paths": 0, "nif_ms": 0.0, "trace_ms": 0.0, "acc_ms": 0.0, "nif_launches": 0}
    barrier()
    t0 = time.perf_counter()
    for step in range(args.steps):
        r.path_trace()
        state["steps_in_interval"] += 1
        st = r.stats()
        agg["escaped"] += st.escaped
        agg["segments"] += st.segments
        agg["paths"] += st.paths
        agg["nif_ms"] += st.nif_ms
        agg["trace_ms"] += st.path_trace_ms
        agg["acc_ms"] += st.accumulate_ms
        agg["nif_launches"] += st.nif_launches
        if args.save_interval > 0 and (step + 1) % args.save_interval == 0 and step + 1 < args.steps:
            hand_off(last=False)
    hand_off(last=True)
    barrier()
    elapsed = time.perf_counter() - t0

    def max_over_ranks(x):
        if not multi:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = max_over_ranks(elapsed)

    # Calibration, same invocation (never `value`): the NIF stage of the last step's largest batch again with nothing beside
    # it -- same kernel, same resident queue (pt_calibrate_nif).  `nif_alone_tflops` moves with the device (the pool's boxes
    # differ by 5-7 %); `value_over_alone` = the rate inside the pipelined step / this rate moves only when the pipeline
    # around the kernel changes.  A slow box lowers both numbers; a regression lowers the ratio.
    calib = None
    if rank == 0 and not (multi and rehearsal):
        try:
            cal_ms, cal_evals = r.calibrate_nif(launches=4)
            calib = {"ms_per_launch": cal_ms, "evaluations_per_launch": int(cal_evals), "launches": 4}
        except ptmi.PtError as e:
            calib = {"error": str(e)}
    kernel_name = r.nif_kernel_name()

    # Second leg, same invocation: the step as the reference times it (PathTracerApp.cpp:692-694,764-767): programs
    # setup -> path_trace -> read_results, i.e. with the worklist's H2D and D2H copies (20 B per item each way) inside.
    # `value` above keeps the inputs resident, as this benchmark's contract requires; this is the reference's own clock.
    ref_work = state["work"].copy()
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        r.setup(ref_work)
        r.path_trace()
        r.read_results(ref_work)
    barrier()
    ref_elapsed = max_over_ranks(time.perf_counter() - t1)

    # Third leg: the trace stage on its own (constant sky, so no NIF kernel runs beside it), one warm-up + one timed step
    # of the same worklist.  Gives the stage's stand-alone time and, with the per-path instruction count of the committed
    # PMC pass, its rate against the VALU issue bound -- the kernel keeps ray state in registers and is VALU-bound.
    r.set_constant_env((1.0, 1.0, 1.0))
    r.path_trace()
    r.path_trace()
    alone = r.stats()

    if rank == 0:
        flops = int(st.nif_flops_per_sample)
        total_samples = W * H * spp * args.steps
        nif_s = agg["nif_ms"] * 1e-3
        achieved = agg["escaped"] * flops / nif_s / 1e12 if nif_s > 0 else 0.0
        wide = args.hidden > 320
        kernel = kernel_name                     # reported by the library (pt_nif_kernel_name): what launch_nif dispatched
        traffic, traffic_src = hbm_traffic_from_profile("nifg16_layer_kernel<0" if wide else "nif_kernel_v3<%d" % args.hidden)
        out = {
            "metric": "Mpath-samples/sec @%dx%d, %d spp/step, depth %d" % (W, H, spp, depth),
            "value": total_samples / elapsed / 1e6,
            "unit": "Mpath-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%d spheres scene, NIF %dx%d fp16 (synthetic weights, urban_alley "
                                   "metadata), %d spp/step, depth %d" % (W, H, args.layers, args.hidden, spp, depth),
                       "parallelism": "image tiles 16x16 round-robin over %d GPU(s), one RCCL gather of HDR tiles"
                                      % world,
                       "trace_dtype": "f32", "nif_flops_per_sample": flops,
                       "escaped_fraction": agg["escaped"] / max(agg["paths"], 1),
                       "segments_per_path": agg["segments"] / max(agg["paths"], 1)},
            "reference_step_definition": {
                "value": total_samples / ref_elapsed / 1e6, "unit": "Mpath-samples/s",
                "ms_per_step": ref_elapsed / args.steps * 1e3, "steps": args.steps,
                "what": "setup -> path_trace -> read_results per step, worklist H2D + D2H inside the clock "
                        "(PathTracerApp.cpp:692-694,764-767); measured in this run"},
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS,
                         "avg_launch_ms": agg["nif_ms"] / max(agg["nif_launches"], 1),
                         "launches": agg["nif_launches"], "traffic": traffic,
                         "traffic_unit": "bytes/launch, NOT measured in this run: PMC FETCH_SIZE x2 + WRITE_SIZE of separate "
                                         "rocprofv3 passes, read from %s" % traffic_src,
                         "rank0_stage_ms": {"trace": agg["trace_ms"], "nif": agg["nif_ms"], "accumulate": agg["acc_ms"]}},
        }
        # Trace stage (ray-gen, intersect, shade, compact; accumulate is its own kernel).  Everything here is measured in
        # this run except the per-path constants, which come from the named PMC file.  Accounting as SURVEY.md 8(d) /
        # BASELINE.md section 4 define it: algorithmic bytes = 96 B per path segment + 88 B per escaped path -- the traffic of
        # a wavefront tracer that keeps ray state in HBM between bounces.  This design keeps it in registers, so the bytes
        # the counters see are a fraction of that figure (avoided, not wasted) and the stage is bound by the vector ALU:
        # ONE fraction is reported, the share of all SIMD cycles in which the VALU is executing.
        pmc, pmc_src = trace_counters_from_profile(depth)
        alone_s = alone.path_trace_ms * 1e-3
        seg_step, esc_step = agg["segments"] / args.steps, agg["escaped"] / args.steps
        ts = {"bound": "valu",
              "standalone_ms_per_step": alone.path_trace_ms, "standalone_accumulate_ms_per_step": alone.accumulate_ms,
              "standalone_Mpath_samples_per_s": alone.paths / max(alone_s, 1e-9) / 1e6,
              "overlapped_ms_per_step": agg["trace_ms"] / args.steps,
              "rays_per_sec": agg["segments"] * world / elapsed,
              "algorithmic_bytes_per_step": 96.0 * seg_step + 88.0 * esc_step,
              "algorithmic_bytes_model": "96 B x path segments + 88 B x escaped paths per step (SURVEY.md 8(d)): assumes ray state "
                                         "travels through HBM between bounces; here it stays in registers",
              "what": "stand-alone = one constant-sky step of the same worklist in this run (no NIF kernel beside it); "
                      "overlapped = the trace kernels' own HIP-event time while the NIF kernel shares the CUs"}
        if pmc:
            if pmc.get("hbm_bytes_per_path"):
                ts["counter_bytes_per_step"] = pmc["hbm_bytes_per_path"] * agg["paths"] / args.steps
                ts["counter_over_algorithmic"] = ts["counter_bytes_per_step"] / max(ts["algorithmic_bytes_per_step"], 1.0)
                ts["achieved_hbm_GBps"] = pmc["hbm_bytes_per_path"] * alone.paths / max(alone_s, 1e-9) / 1e9
                ts["hbm_peak_GBps"] = HBM_PEAK_GBPS
            ts["frac"] = pmc.get("valu_busy_fraction")
            ts["frac_is"] = "VALU busy: cycles the SIMDs spend executing vector instructions / all SIMD cycles of the launch"
            ts["valu_wave_instr_per_path"] = pmc["valu_wave_instr_per_path"]
            ts["cycles_per_valu_wave_instr"] = pmc.get("cycles_per_valu_wave_instr")
            ts["lane_utilisation"] = pmc.get("lane_utilisation")
            ts["G_wave_instr_per_s"] = pmc["valu_wave_instr_per_path"] * alone.paths / max(alone_s, 1e-9) / 1e9
            ts["counters_from"] = "%s (not measured in this run)" % pmc_src
        out["trace_stage"] = ts
        if calib and "error" not in calib:
            alone_tf = calib["evaluations_per_launch"] * flops / (calib["ms_per_launch"] * 1e-3) / 1e12
            out["roofline"]["nif_alone_tflops"] = alone_tf
            out["roofline"]["nif_alone_frac"] = alone_tf / MFMA_F16_DENSE_PEAK_TFLOPS
            out["roofline"]["value_over_alone"] = achieved / alone_tf if alone_tf > 0 else None
            out["roofline"]["nif_alone"] = dict(calib, what="pt_calibrate_nif: the NIF stage of the last step's largest batch re-run "
                                                "alone on the device right after the timed steps (1 untimed + 4 timed launches)")
            if traffic and not wide:
                # algorithmic bytes of the launch the PMC figure describes (a full-size batch): 24 B of queue entry read +
                # 12 B of radiance written per evaluation
                out["roofline"]["algorithmic_bytes_per_launch"] = 36 * calib["evaluations_per_launch"]
                out["roofline"]["traffic_over_algorithmic"] = traffic / (36.0 * calib["evaluations_per_launch"])
        elif calib:
            out["roofline"]["nif_alone"] = calib
        try:
            out["runtime"] = dict(ptmi.runtime_info(), torch=torch.__version__, torch_hip=getattr(torch.version, "hip", None),
                                  torch_imported_first=True,
                                  what="the shared objects libptmi.so's HIP / RCCL imports are bound to in this process "
                                       "(pt_runtime_info): torch is imported first, so they are the copies PyTorch ships")
        except Exception as e:   # noqa: BLE001
            out["runtime"] = {"error": str(e)}
        if world == 1 and not args.no_secondary and not wide and not dist_mode:
            out["secondary"] = secondary_configs(ptmi, nif_assets, W, H, depth, meta, mean)
        if world == 1 and not args.no_cpu_baseline and not dist_mode:
            # the GPU's rate on configs[0] first: the CPU legs start 100+ OpenMP threads that keep spinning afterwards
            try:
                c1_gpu = c1_on_the_gpu(ptmi)
            except Exception as e:   # noqa: BLE001 -- never cost the headline line
                c1_gpu = {"gpu_error": str(e)}
            try:
                out["cpu_baseline"] = cpu_baseline(W, H, depth, layers, meta, mean, c1_gpu)
            except Exception as e:   # noqa: BLE001 -- e.g. the timing build does not compile on this host (oracle/Makefile needs AVX2 + F16C + FMA)
                out["cpu_baseline"] = {"error": str(e), "c1": c1_gpu}
        if multi or args.dump_film:
            if args.save_interval > 0:
                film = (film_sum / args.steps).astype(np.float32)
                out["config"]["save_interval"] = args.save_interval
                out["config"]["load_balancing"] = bool(redeal)
            else:
                film = partition.assemble_hdr(W, H, world, gathered["tiles"])
            out["config"]["hdr_gather"] = gathered["via"]
            out["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "(unset)")
            out["config"]["film_mean"] = float(film.mean())
            out["config"]["film_nonzero_fraction"] = float((film.sum(axis=2) > 0).mean())
            if args.dump_film:
                np.save(args.dump_film, film)
        if rehearsal:
            out["data"] = "synthetic (REHEARSAL on one shared GPU -- not a measurement)"
        print(json.dumps(out), flush=True)
    r.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
