"""MI355X-native hot path of markp-gc/ipu_path_trace: the per-pixel Monte-Carlo sampling loop and
the NIF environment-light MLP as HIP kernels behind the C-ABI of include/ptmi.h.

`ipu_path_trace_amd.ptmi` is the ctypes binding of libptmi.so.  There is no CPU fallback: importing
works anywhere, but creating a renderer without the built library or without a GPU raises.
"""
from . import nif_assets  # noqa: F401
from .build import build_library, library_path  # noqa: F401
