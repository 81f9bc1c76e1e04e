"""The C-ABI library loads and exports every symbol include/ptmi.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ptmi_lib):
    lib = ptmi_lib.load_library()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libptmi.so does not export %s" % n
    assert sorted(ptmi_lib.EXPORTS) == names
    from ipu_path_trace_amd import ptmi as _binding
    assert lib.pt_abi_version() == _binding.ABI_VERSION == 5   # 5: pt_runtime_info; 2: communicator + pt_gather_hdr, PT_DTYPE_F32; 3: communicator deadlines + pt_comm_abort, pt_tile_costs; 4: pt_nif_kernel_name, pt_calibrate_nif


def test_struct_layouts_match_header(ptmi_lib):
    assert C.sizeof(ptmi_lib.Config) == 56  # 12 x 4 bytes + pointer
    assert C.sizeof(ptmi_lib.Layer) == 32
    assert C.sizeof(ptmi_lib.Stats) == 80
    assert ptmi_lib.TRACE_DTYPE.itemsize == 20 and ptmi_lib.PATH_DTYPE.itemsize == 48


def test_create_rejects_bad_config_and_reports_no_device(ptmi_lib):
    import torch
    lib = ptmi_lib.load_library()
    cfg = ptmi_lib.Config()
    h = C.c_void_p()
    assert lib.pt_create(C.byref(cfg), C.byref(h)) == -1  # struct_size mismatch
    assert b"struct_size" in lib.pt_last_error(None)
    with pytest.raises(ptmi_lib.PtError) as e:
        ptmi_lib.Renderer(0, 10)
    assert e.value.code == -1
    with pytest.raises(ptmi_lib.PtError):
        ptmi_lib.Renderer(64, 64, roulette_depth=0)  # undefined behaviour in the reference (codelets.cpp:221)
    with pytest.raises(ptmi_lib.PtError):
        ptmi_lib.Renderer(64, 64, max_path_length=65)
    if not torch.cuda.is_available():
        # the product path fails loudly without a GPU: there is no CPU fallback behind the C-ABI
        with pytest.raises(ptmi_lib.PtError) as e:
            ptmi_lib.Renderer(64, 64)
        assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_null_handle_calls_fail_cleanly(ptmi_lib):
    lib = ptmi_lib.load_library()
    assert lib.pt_path_trace(None) == -1
    assert lib.pt_destroy(None) == 0
    assert lib.pt_setup(None, None, 0) == -1


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ipu_path_trace_amd/ or bench.py's GPU leg may use it."""
    pkg = os.path.join(ROOT, "ipu_path_trace_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")) or f == "Makefile":
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "pt_oracle" not in src and "libpt_oracle" not in src, os.path.join(d, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)


def test_product_library_carries_no_test_hook(ptmi_lib):
    """Nothing in the environment can change what libptmi.so does: no PTMI_* string (the A/B switches, fault injection and
    tuning knobs live only in libptmi_diag.so, -DPTMI_DIAG_BUILD) and no pt_diag_* symbol; the package never reads an
    environment variable to pick a library; bench.py loads the product library only."""
    lib = ptmi_lib.load_library()
    blob = open(lib._name, "rb").read()
    assert b"PTMI_" not in blob
    assert not hasattr(lib, "pt_diag_inject_fault") and not hasattr(lib, "pt_diag_stamps")
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib._name], capture_output=True, text=True).stdout
    assert "getenv" not in syms, "libptmi.so must not read the environment"
    for rel in ("ipu_path_trace_amd/build.py", "ipu_path_trace_amd/ptmi.py", "bench.py"):
        src = open(os.path.join(ROOT, rel)).read()
        assert "PTMI_LIBRARY" not in src, rel
        if rel == "bench.py":
            assert "diag" not in src, "bench.py measures the product library only"
    diag = ptmi_lib.load_library(diag=True)
    assert hasattr(diag, "pt_diag_inject_fault") and hasattr(diag, "pt_diag_stamps")


def test_comm_entry_points_reject_bad_arguments_without_a_gpu(ptmi_lib):
    """librccl is linked into libptmi.so (the north_star's RCCL gather lives behind the C-ABI); argument checks need
    no device."""
    lib = ptmi_lib.load_library()
    assert lib.pt_comm_get_unique_id(None) == -1
    assert lib.pt_comm_init_rank(None, None, 0, 1) == -1
    assert lib.pt_comm_init_all(None, 0) == -1
    assert lib.pt_gather_hdr(None, 0, 10, None) == -1
    assert lib.pt_comm_set_timeout(None, 1000) == -1 and lib.pt_comm_abort(None) == -1
    assert lib.pt_film_accumulate(None) == -1
    import subprocess
    deps = subprocess.run(["readelf", "-d", lib._name], capture_output=True, text=True).stdout
    assert "librccl.so" in deps
