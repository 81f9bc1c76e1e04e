"""The machinery that keeps a stuck RCCL call from hanging a caller (ipu_path_trace_amd/csrc/ptmi_comm_worker.h), on the CPU
under ThreadSanitizer.  RCCL 2.27.7 stays inside ncclCommInitRankConfig / ncclCommAbort for as long as a peer is missing,
whatever ncclConfig_t::blocking says, so pt_comm_init_* and pt_gather_hdr run every RCCL call on a
long-lived worker thread and wait for it against the handle's deadline.  The reference has no counterpart: the shards of
one Poplar engine cannot lose each other (src/PathTracerApp.cpp:205-252).  The real thing is exercised on the GPU by
tests/test_multi_gpu_launch.py (both RCCL bindings)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_bounded_worker_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "comm_worker")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer",
                            "-I" + os.path.join(ROOT, "ipu_path_trace_amd", "csrc"), "-o", exe,
                            os.path.join(ROOT, "tests", "comm_worker_main.cpp"), "-lpthread"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0")   # parked / stuck workers are leaked on purpose
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "COMM_WORKER_OK" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr, run.stderr[-4000:]
