"""Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC"]


def library_path():
    # PTMI_LIBRARY selects another in-tree build of the same sources (e.g. the -DPTMI_DIAG_BUILD ablation library)
    return os.environ.get("PTMI_LIBRARY") or os.path.join(_PKG, "libptmi.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> ipu_path_trace_amd/libptmi.so"""
    srcs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC))]
    srcs.append(os.path.join(_ROOT, "include", "ptmi.h"))
    out = library_path()
    if not force and _newer(out, srcs):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # librccl is linked directly: the HDR-tile gather (pt_gather_hdr) is part of the product boundary
    cmd = [hipcc] + HIPCC_FLAGS + ["-I" + os.path.join(_ROOT, "include"), "-o", out, os.path.join(_CSRC, "ptmi.hip"),
                                   "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out
