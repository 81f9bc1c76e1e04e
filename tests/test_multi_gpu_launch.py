"""C4 (image tile-partitioned over N GPUs, one gather of HDR tiles): the N > 1 path must be STARTABLE the way the driver
starts it (`python bench.py --gpus N`, no launcher around it) and UNABLE TO HANG.  Reference: `--ipus N` is one command
(src/main.cpp:17-19) and the shards of one Poplar engine cannot lose each other (src/PathTracerApp.cpp:205-252); here
the ranks are processes or threads around an RCCL communicator, so every wait has a deadline.

What a one-GPU box can show: the launch path end to end (two ranks sharing the GPU, gloo in place of RCCL), and the
deadline / abort machinery against a peer that never arrives (a communicator of world size 2 whose rank 1 is never
started).  A real two-rank RCCL exchange needs two GPUs (RCCL refuses two ranks on one device): unmeasured, DESIGN.md 6.
"""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_REHEARSAL")}
    env.update(extra)
    return env


def test_bench_spawns_its_own_ranks_and_relays_their_failure_without_a_gpu():
    """No GPU here: the two child ranks refuse to run, and the parent (which never touched the GPU) relays their exit
    code instead of the old "launch with torch.distributed.run" refusal.  On a GPU box the same command renders."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_two_ranks_started_by_bench_itself")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--samples-per-step", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(BENCH_REHEARSAL="1"), cwd=ROOT)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr, p.stderr[-1500:]
    assert "launch with torch.distributed.run" not in p.stderr


def test_bench_dist_switch_spawns_a_rank_even_for_one_gpu_without_a_gpu():
    """`--gpus 1 --dist` (and BENCH_FORCE_DIST=1) must go through spawn_ranks(1): here, without a GPU, the one child rank refuses
    to run and the parent relays its exit code; plain `--gpus 1` refuses in-process.  (On a GPU box:
    test_bench_takes_the_real_n_rank_branch_at_world_size_one.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_takes_the_real_n_rank_branch_at_world_size_one")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--samples-per-step", "1",
            "--no-cpu-baseline", "--no-secondary"]
    for extra, env in ((["--dist"], {}), ([], {"BENCH_FORCE_DIST": "1"})):
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=300, env=_clean_env(**env), cwd=ROOT)
        assert p.returncode != 0 and "needs an MI355X" in p.stderr, p.stderr[-1500:]
        assert "torch.distributed" in p.stderr or "ChildFailedError" in p.stderr or "elastic" in p.stderr, p.stderr[-1500:]   # it WAS a launched rank
    p = subprocess.run(base, capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert p.returncode != 0 and "needs an MI355X" in p.stderr and "elastic" not in p.stderr


@pytest.mark.gpu
def test_bench_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: rc 0 and a bench line for two ranks whose film covers the image."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--samples-per-step", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                       env=_clean_env(BENCH_REHEARSAL="1"), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["value"] > 0
    assert out["config"]["film_nonzero_fraction"] > 0.99          # both ranks' tiles reached rank 0's film
    assert "REHEARSAL" in out["data"]                              # and nobody can take it for a measurement


@pytest.mark.gpu
def test_bench_rehearsal_of_c4_four_ranks_film_equals_one_rank(tmp_path):
    """BASELINE config C4's job shape through bench.py itself, started the way the driver starts it: N ranks, a save
    interval, tiles re-dealt between the ranks by measured path length, one gather of HDR tiles per interval.  The film rank 0
    assembles must equal the one-rank film of the same sample indices BIT FOR BIT (RNG keyed by pixel and absolute sample
    index; per-pixel fp32 sums in iteration order whichever rank owns the pixel).  Four ranks, not C4's eight: every rank
    is a process holding GPU 0, and the pool's GPU boxes allow six processes on the card at once, this test process
    included (the driver's own 8-GPU node runs one rank per GPU).  World size 8 itself is covered where one process can do
    it -- tests/test_host.py::test_cli_multi_device_loop_on_one_gpu (ipu_trace --ipus 8 --devices 0 x 8) -- and on the CPU
    (tests/test_partition.py, 8 gloo ranks)."""
    import numpy as np
    common = ["--steps", "2", "--warmup", "1", "--samples-per-step", "8", "--save-interval", "1", "--no-cpu-baseline",
              "--no-secondary", "--width", "368", "--height", "272"]
    films = {}
    for n, extra in ((1, []), (4, ["--enable-load-balancing"])):
        path = str(tmp_path / ("film%d.npy" % n))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dump-film", path] + common + extra,
                           capture_output=True, text=True, timeout=900, env=_clean_env(BENCH_REHEARSAL="1"), cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["steps"] == 2
        assert out["config"]["film_nonzero_fraction"] > 0.99
        if n > 1:
            assert out["config"]["load_balancing"] is True and out["config"]["save_interval"] == 1
            assert "HSA_ENABLE_IPC_MODE_LEGACY" in out["config"]         # which IPC mode the ranks ran with is on record
            assert "REHEARSAL" in out["data"]
        films[n] = np.load(path)
    assert films[4].shape == (272, 368, 3) and films[4].tobytes() == films[1].tobytes()


_LONE_RANK = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    BINDING = %r
    if BINDING == "no_torch":
        sys.modules["torch"] = None          # `import torch` now raises ImportError: ptmi.load_library loads libptmi.so without it
    elif BINDING == "torch_first":
        import torch
    from ipu_path_trace_amd import ptmi
    info = ptmi.runtime_info()
    print("RUNTIME", info)
    if BINDING == "no_torch":
        assert "torch" not in sys.modules or sys.modules["torch"] is None
        assert "/opt/rocm" in info["librccl"] and "/opt/rocm" in info["libamdhip64"], info
        assert info["rccl_version"] == info["rccl_compiled"], info      # the RCCL libptmi.so was compiled against
    elif BINDING == "torch_first":
        assert "/torch/lib/" in info["librccl"] and "/torch/lib/" in info["libamdhip64"], info
    r = ptmi.Renderer(32, 32, max_path_length=4)
    r.comm_set_timeout(4000)
    uid = ptmi.comm_unique_id()
    t = time.time()
    try:
        r.comm_init_rank(uid, 0, 2)          # rank 1 is never started
        print("UNEXPECTED: set-up of a 2-rank communicator finished with one rank")
        sys.exit(3)
    except ptmi.PtError as e:
        dt = time.time() - t
        print("CODE", e.code, "AFTER %%.1f" %% dt, "MSG", e)
        assert e.code == -7 and 3.0 < dt < 60.0, (e.code, dt)
    # the handle is still good for everything that needs no peer, and takes a new communicator
    r.set_constant_env((1.0, 1.0, 1.0))
    r.init_render_settings(samples_per_step=2)
    rec = ptmi.worklist(32, 32)
    r.setup(rec)
    r.path_trace()
    try:
        r.gather_hdr(32 * 32)
        print("UNEXPECTED: gather on an aborted communicator")
        sys.exit(4)
    except ptmi.PtError as e:
        assert e.code == -7 and "aborted" in str(e), str(e)
    r.comm_init_rank(ptmi.comm_unique_id(), 0, 1)
    tiles = r.gather_hdr(32 * 32)
    assert tiles.shape == (1, 1024, 3) and tiles.max() > 0
    r.close()
    print("LONE_RANK_OK")
""")


@pytest.mark.gpu
def test_a_rank_whose_peer_never_arrives_times_out_and_recovers(tmp_path):
    """Real RCCL, world size 2, one rank: the non-blocking set-up is polled against pt_comm_set_timeout's deadline,
    the communicator is aborted, PT_ERR_COMM comes back -- and the process is still alive and usable.  Run in a child
    with a hard limit so that a regression (a hang) fails this test instead of stalling the suite."""
    script = tmp_path / "lone_rank.py"
    script.write_text(_LONE_RANK % (ROOT, "default"))
    t = time.time()
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=_clean_env(), cwd=ROOT)
    assert p.returncode == 0 and "LONE_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    assert time.time() - t < 200


@pytest.mark.gpu
@pytest.mark.parametrize("binding", ["torch_first", "no_torch"])
def test_both_runtime_bindings_pass_the_rccl_deadline_and_gather_tests(tmp_path, binding):
    """libptmi.so imports librccl.so.1 / libamdhip64.so.7 by SONAME, so a process that imported torch first runs the library
    on the copies PyTorch ships (an older RCCL than the headers it was compiled against: pt_comm_init_rank stamps the config
    with the running library's version), and a process without torch -- the C++ CLI ipu_trace, or this child with torch
    hidden -- runs it on ROCm's own.  bench.py is the first kind at every N.  Both bindings must pass the same body: the
    lone-rank deadline on a world-2 communicator, recovery, and the single-rank gather on a real non-blocking communicator;
    pt_runtime_info says which is which."""
    script = tmp_path / ("lone_rank_%s.py" % binding)
    script.write_text(_LONE_RANK % (ROOT, binding))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert p.returncode == 0 and "LONE_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    assert "RUNTIME" in p.stdout


def _bench_line(args, timeout=900, **env):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=_clean_env(**env), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_takes_the_real_n_rank_branch_at_world_size_one(tmp_path):
    """`bench.py --gpus 1 --dist`: everything an N-rank run does that one GPU can execute, with the REAL backend -- the child
    started by spawn_ranks(1), init_process_group("nccl"), the ncclUniqueId broadcast, pt_comm_init_rank(id, 0, 1), the
    all-reduced agreement on a CUDA tensor, the warm-up product gather, max over ranks on CUDA, calibrate-then-barrier.  The
    film that went through the communicator equals the plain one-GPU film bit for bit, and the line names the product gather
    and the runtime it ran on.  Second case: a damaged unique id -- every rank (the one there is) must agree on the fallback
    to torch.distributed's gather, say so in the line, and still deliver the same film.
    Reference: --ipus N runs as one command (src/main.cpp:17-19, src/PathTracerApp.cpp:205-252)."""
    import numpy as np
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--samples-per-step", "8", "--no-cpu-baseline", "--no-secondary",
              "--width", "368", "--height", "272"]
    plain = str(tmp_path / "plain.npy")
    out = _bench_line(common + ["--dump-film", plain])
    assert out["config"]["hdr_gather"] == "pt_gather_hdr of one tile (no communicator)"
    assert "/torch/lib/" in out["runtime"]["librccl"] and out["runtime"]["rccl_version"] > 0

    dist_film = str(tmp_path / "dist.npy")
    out = _bench_line(common + ["--dist", "--dump-film", dist_film])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["hdr_gather"] == "pt_gather_hdr (RCCL inside libptmi.so)", out["config"]
    assert out["runtime"]["torch_imported_first"] and "/torch/lib/" in out["runtime"]["librccl"]
    assert "nif_alone" in out["roofline"]                                     # calibrate-then-barrier ran
    assert np.load(dist_film).tobytes() == np.load(plain).tobytes()

    env_film = str(tmp_path / "env.npy")
    out = _bench_line(common + ["--dump-film", env_film], BENCH_FORCE_DIST="1")   # the environment switch does the same
    assert out["config"]["hdr_gather"] == "pt_gather_hdr (RCCL inside libptmi.so)"
    assert np.load(env_film).tobytes() == np.load(plain).tobytes()

    # C4's job shape on the same branch: a save interval, tiles re-dealt by measured path length (the all-reduce of the tile
    # costs on a CUDA tensor, pt_tile_costs, pt_setup of the new worklist), one product gather per interval
    shaped, shaped_plain = str(tmp_path / "shaped.npy"), str(tmp_path / "shaped_plain.npy")
    interval = ["--save-interval", "1"]
    out = _bench_line(common + interval + ["--dump-film", shaped_plain])
    out = _bench_line(common + interval + ["--dist", "--enable-load-balancing", "--dump-film", shaped])
    assert out["config"]["hdr_gather"] == "pt_gather_hdr (RCCL inside libptmi.so)" and out["config"]["load_balancing"] is True
    assert np.load(shaped).tobytes() == np.load(shaped_plain).tobytes()

    fallback = str(tmp_path / "fallback.npy")
    out = _bench_line(common + ["--dist", "--comm-fault", "corrupt-id", "--comm-timeout-ms", "8000", "--dump-film", fallback])
    assert out["config"]["hdr_gather"].startswith("torch.distributed gather (pt_comm_init_rank failed"), out["config"]
    assert np.load(fallback).tobytes() == np.load(plain).tobytes()
