"""Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU).

Two libraries come out of the same sources:
  libptmi.so       the product: no environment variable, test hook or ablation kernel is compiled into it
  libptmi_diag.so  -DPTMI_DIAG_BUILD: the profiling / test build (kernel variants of csrc/diag/, A/B switches read
                   from the environment, in-kernel clock stamps, pt_diag_* entry points incl. fault injection).
                   Only tests/ and scripts/ load it, always by explicit path.
"""
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC"]


def library_path(diag=False):
    return os.path.join(_PKG, "libptmi_diag.so" if diag else "libptmi.so")


def _sources():
    out = []
    for d, _, files in os.walk(_CSRC):
        out += [os.path.join(d, f) for f in sorted(files)]
    out.append(os.path.join(_ROOT, "include", "ptmi.h"))
    return out


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_library(force=False, verbose=False, diag=False):
    """hipcc --offload-arch=gfx950 ... -> ipu_path_trace_amd/libptmi.so (diag=True: libptmi_diag.so)"""
    out = library_path(diag)
    if not force and _newer(out, _sources()):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # librccl is linked directly: the HDR-tile gather (pt_gather_hdr) is part of the product boundary
    cmd = [hipcc] + HIPCC_FLAGS + (["-DPTMI_DIAG_BUILD"] if diag else []) + [
        "-I" + os.path.join(_ROOT, "include"), "-I" + _CSRC, "-o", out, os.path.join(_CSRC, "ptmi.hip"),
        "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out
