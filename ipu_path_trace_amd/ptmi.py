"""ctypes binding of include/ptmi.h (libptmi.so) plus a thin Renderer that sequences the calls the
way PathTracerApp::execute does (reference: src/PathTracerApp.cpp:612-614, :692-694)."""
import ctypes as C
import os

import numpy as np

from .build import library_path

TRACE_DTYPE = np.dtype([("u", "<u2"), ("v", "<u2"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"),
                        ("sampleCount", "<u2"), ("pathLength", "<u2")], align=True)
assert TRACE_DTYPE.itemsize == 20  # src/codelets/TraceRecord.hpp:7-19

PATH_DTYPE = np.dtype([("length", "<u4"), ("escaped", "<u4"), ("dir", "<f4", 3), ("uv", "<f4", 2),
                       ("throughput", "<f4", 3), ("cam", "<f4", 2)])
assert PATH_DTYPE.itemsize == 48

AA_NORMAL, AA_UNIFORM, AA_TRUNCATED_NORMAL = 0, 1, 2
SAMPLES_HALF, SAMPLES_FLOAT = 0, 1
DTYPE_F16, DTYPE_F32 = 0, 1

ABI_VERSION = 5          # PTMI_ABI_VERSION of include/ptmi.h this binding was written against
EXPORTS = ["pt_abi_version", "pt_create", "pt_destroy", "pt_last_error", "pt_upload_nif", "pt_set_constant_env",
           "pt_set_render_settings", "pt_setup", "pt_path_trace", "pt_read_results", "pt_get_stats",
           "pt_export_hdr_device", "pt_clear_accumulators", "pt_synchronize", "pt_nif_infer", "pt_trace_paths",
           "pt_comm_get_unique_id", "pt_comm_init_rank", "pt_comm_init_all", "pt_comm_info", "pt_comm_set_timeout", "pt_comm_abort",
           "pt_gather_hdr", "pt_film_accumulate", "pt_tile_costs_enable", "pt_tile_costs", "pt_film_seed",
           "pt_nif_kernel_name", "pt_calibrate_nif", "pt_runtime_info"]
COMM_ID_BYTES = 128
HDR_ACCUMULATORS, HDR_FILM = 0, 1


class PtError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (src/ipu_utils.hpp:532-535)."""

    def __init__(self, code, message):
        super().__init__("ptmi error %d: %s" % (code, message))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("max_path_length", C.c_uint32), ("roulette_depth", C.c_uint32), ("stop_prob", C.c_float),
                ("refractive_index", C.c_float), ("aa_noise_type", C.c_int32), ("sample_precision", C.c_int32),
                ("device", C.c_int32), ("max_work_items", C.c_uint32), ("iterations_per_batch", C.c_uint32),
                ("stream", C.c_void_p)]


class Layer(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("kernel", C.c_void_p), ("bias", C.c_void_p),
                ("dtype", C.c_int32), ("relu", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("escaped", C.c_uint64),
                ("nif_flops_per_sample", C.c_uint64), ("path_trace_ms", C.c_double), ("nif_ms", C.c_double),
                ("accumulate_ms", C.c_double), ("total_ms", C.c_double), ("trace_launches", C.c_uint32),
                ("nif_launches", C.c_uint32), ("accumulate_launches", C.c_uint32), ("first_sample", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_libs = {}


def load_library(diag=False):
    """Load libptmi.so (the product).  Raises if it has not been built: the product path has no fallback.
    diag=True loads libptmi_diag.so, the profiling / test build (only tests/ and scripts/ ask for it)."""
    if diag in _libs:
        return _libs[diag]
    path = library_path(diag)
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')" % path)
    try:
        # torch bundles its own HIP runtime: when both live in one process it has to be loaded first, or torch later
        # finds "No HIP GPUs" (two copies of libamdhip64 with one SONAME).  No torch, no problem.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.pt_abi_version.restype = C.c_int
    if L.pt_abi_version() != ABI_VERSION:   # a stale in-tree build: the structs below would not match the library's
        raise RuntimeError("%s has ABI version %d, this binding needs %d: rebuild it (python -c 'import __graft_entry__ as g; "
                           "g.build()')" % (path, L.pt_abi_version(), ABI_VERSION))
    L.pt_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.pt_destroy.argtypes = [C.c_void_p]
    L.pt_last_error.restype = C.c_char_p
    L.pt_last_error.argtypes = [C.c_void_p]
    L.pt_upload_nif.argtypes = [C.c_void_p, C.POINTER(Layer), C.c_uint32, C.c_uint32, C.c_float,
                                C.POINTER(C.c_float), C.c_int32]
    L.pt_set_constant_env.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.pt_set_render_settings.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_uint32]
    L.pt_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pt_path_trace.argtypes = [C.c_void_p]
    L.pt_read_results.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Stats)]
    L.pt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.pt_export_hdr_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pt_comm_get_unique_id.argtypes = [C.c_void_p]
    L.pt_comm_init_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.pt_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.pt_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pt_comm_set_timeout.argtypes = [C.c_void_p, C.c_uint32]
    L.pt_comm_abort.argtypes = [C.c_void_p]
    L.pt_gather_hdr.argtypes = [C.c_void_p, C.c_int32, C.c_size_t, C.c_void_p]
    L.pt_film_accumulate.argtypes = [C.c_void_p]
    L.pt_film_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pt_tile_costs_enable.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.pt_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pt_clear_accumulators.argtypes = [C.c_void_p]
    L.pt_synchronize.argtypes = [C.c_void_p]
    L.pt_nif_infer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.pt_trace_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.pt_nif_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.pt_calibrate_nif.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.pt_runtime_info.argtypes = [C.c_char_p, C.c_size_t]
    if diag:
        L.pt_diag_inject_fault.argtypes = [C.c_void_p, C.c_int32]
        L.pt_diag_stamps.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_diag_nif_clock.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_diag_comm_self_exchange.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    _libs[diag] = L
    return L


def degrees_to_radians_f32(deg):
    """PathTracerApp.cpp:574: float fov = deg * (M_PI / 180.f) evaluated in single precision."""
    return float(np.float32(deg) * np.float32(np.pi / 180.0))


def rotation_to_radians_f32(deg):
    """PathTracerApp.cpp:584: (degrees / 360.f) * (2.0 * M_PI)."""
    return float(np.float32((np.float32(deg) / np.float32(360.0)) * (2.0 * np.pi)))


class Renderer:
    """One device context (pt_handle).  Methods are named after the reference's Poplar programs."""

    def __init__(self, width, height, max_work_items=None, max_path_length=10, roulette_depth=3, stop_prob=0.3,
                 refractive_index=1.5, aa_noise_type=AA_NORMAL, sample_precision=SAMPLES_HALF, device=0,
                 iterations_per_batch=0, stream=None, diag=False):
        self._lib = load_library(diag)
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.width, cfg.height = width, height
        cfg.max_path_length, cfg.roulette_depth = max_path_length, roulette_depth
        cfg.stop_prob, cfg.refractive_index = stop_prob, refractive_index
        cfg.aa_noise_type, cfg.sample_precision = aa_noise_type, sample_precision
        cfg.device = device
        cfg.max_work_items = max_work_items if max_work_items else width * height
        cfg.iterations_per_batch = iterations_per_batch
        cfg.stream = stream
        self.handle = C.c_void_p()
        rc = self._lib.pt_create(C.byref(cfg), C.byref(self.handle))
        if rc:
            msg = self._lib.pt_last_error(None).decode()
            self.handle = None
            raise PtError(rc, msg)
        self._keep = []

    def _check(self, rc):
        if rc:
            raise PtError(rc, self._lib.pt_last_error(self.handle).decode())

    def close(self):
        if getattr(self, "handle", None):
            self._lib.pt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    # ---- program "init_nif_weights"
    def init_nif_weights(self, layers, embedding_dim, max_value, mean_folded, log_tonemap=True):
        arr = (Layer * len(layers))()
        keep = []
        for i, (k, b, relu) in enumerate(layers):
            # float32 kernels are handed over as float32 (the library rounds them to binary16, as documented in
            # include/ptmi.h); everything else is passed as float16, the type the reference's shipped NIFs use
            dt = np.float32 if np.asarray(k).dtype == np.float32 else np.float16
            k = np.ascontiguousarray(k, dtype=dt)
            keep.append(k)
            arr[i].rows, arr[i].cols = k.shape
            arr[i].kernel = k.ctypes.data
            if b is not None:
                b = np.ascontiguousarray(b, dtype=dt)
                keep.append(b)
                arr[i].bias = b.ctypes.data
            arr[i].dtype = DTYPE_F32 if dt == np.float32 else DTYPE_F16
            arr[i].relu = int(bool(relu))
        mean = (C.c_float * 3)(*[float(x) for x in mean_folded])
        self._check(self._lib.pt_upload_nif(self.handle, arr, len(layers), embedding_dim, float(max_value), mean,
                                            int(log_tonemap)))

    def set_constant_env(self, rgb):
        v = (C.c_float * 3)(*[float(x) for x in rgb])
        self._check(self._lib.pt_set_constant_env(self.handle, v))

    # ---- program "init_render_settings"
    def init_render_settings(self, seed=1, aa_noise_scale=0.3, fov_degrees=90.0, env_rotation_degrees=0.0,
                             samples_per_step=1):
        self._check(self._lib.pt_set_render_settings(self.handle, seed, aa_noise_scale,
                                                     degrees_to_radians_f32(fov_degrees),
                                                     rotation_to_radians_f32(env_rotation_degrees), samples_per_step))

    # ---- programs "setup" / "path_trace" / "read_results"
    def setup(self, records):
        assert records.dtype == TRACE_DTYPE and records.flags.c_contiguous
        self._check(self._lib.pt_setup(self.handle, records.ctypes.data, records.size))

    def path_trace(self):
        self._check(self._lib.pt_path_trace(self.handle))

    def read_results(self, records):
        assert records.dtype == TRACE_DTYPE and records.flags.c_contiguous
        st = Stats()
        self._check(self._lib.pt_read_results(self.handle, records.ctypes.data, records.size, C.byref(st)))
        return st

    def stats(self):
        st = Stats()
        self._check(self._lib.pt_get_stats(self.handle, C.byref(st)))
        return st

    def nif_kernel_name(self):
        """The NIF kernel(s) the library dispatched at its last NIF launch ('' before the first)."""
        buf = C.create_string_buffer(512)
        self._check(self._lib.pt_nif_kernel_name(self.handle, buf, len(buf)))
        return buf.value.decode()

    def calibrate_nif(self, launches=4):
        """The NIF stage of the last path_trace's largest batch again, alone on the device.
        Returns (milliseconds per launch, NIF evaluations per launch)."""
        ms, evals = C.c_double(), C.c_uint64()
        self._check(self._lib.pt_calibrate_nif(self.handle, launches, C.byref(ms), C.byref(evals)))
        return ms.value, evals.value

    def export_hdr_device(self, device_ptr, n):
        self._check(self._lib.pt_export_hdr_device(self.handle, C.c_void_p(device_ptr), n))

    # ---- multi-GPU film hand-off (RCCL inside libptmi.so)
    def comm_init_rank(self, unique_id, rank, world):
        """Join the RCCL communicator made from `unique_id` (bytes from `comm_unique_id()` on rank 0)."""
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.pt_comm_init_rank(self.handle, buf, rank, world))

    def comm_info(self):
        rank, world = C.c_int(), C.c_int()
        self._check(self._lib.pt_comm_info(self.handle, C.byref(rank), C.byref(world)))
        return rank.value, world.value

    def comm_set_timeout(self, milliseconds):
        """Deadline of every communicator operation of this handle (set-up, slot agreement, gather)."""
        self._check(self._lib.pt_comm_set_timeout(self.handle, int(milliseconds)))

    def comm_abort(self):
        """Ask the handle to abort its communicator; callable from another thread while a gather is waiting."""
        self._check(self._lib.pt_comm_abort(self.handle))

    def film_accumulate(self):
        """AccumulatedImage::accumulate + clearInactiveAccumulators on the device (the film stays resident)."""
        self._check(self._lib.pt_film_accumulate(self.handle))

    def film_seed(self, bgr):
        """Set the resident film of the current work items (float32 [n, 3] BGR running sums): the film follows its pixels."""
        bgr = np.ascontiguousarray(bgr, dtype=np.float32)
        self._check(self._lib.pt_film_seed(self.handle, bgr.ctypes.data, bgr.shape[0]))

    def tile_costs_enable(self, tile_w, tile_h):
        """Keep per-tile sums of pathLength on the device (what the path-length balancer needs: kilobytes, not the worklist)."""
        self._tile_grid = (tile_w, tile_h)
        self._check(self._lib.pt_tile_costs_enable(self.handle, tile_w, tile_h))

    def tile_costs(self, width, height):
        """uint64 [tiles]: per-tile sum of pathLength since the last setup (row-major grid of the enabled tile size)."""
        tw, th = self._tile_grid
        n = ((width + tw - 1) // tw) * ((height + th - 1) // th)
        out = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.pt_tile_costs(self.handle, out.ctypes.data, n))
        return out

    def gather_hdr(self, slot_items, source=HDR_ACCUMULATORS):
        """One RCCL gather of HDR tiles to rank 0.  Returns float32 [world, slot_items, 3] (BGR) on rank 0, None elsewhere."""
        rank, world = self.comm_info()
        out = np.empty((world, slot_items, 3), dtype=np.float32) if rank == 0 else None
        self._check(self._lib.pt_gather_hdr(self.handle, source, slot_items, out.ctypes.data if out is not None else None))
        return out

    def clear_accumulators(self):
        self._check(self._lib.pt_clear_accumulators(self.handle))

    def synchronize(self):
        self._check(self._lib.pt_synchronize(self.handle))

    # ---- kernel-level entry points
    def nif_infer(self, u, v):
        u = np.ascontiguousarray(u, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.empty((u.size, 3), dtype=np.float32)
        self._check(self._lib.pt_nif_infer(self.handle, u.ctypes.data, v.ctypes.data, u.size, out.ctypes.data))
        return out

    def trace_paths(self, u, v, sample_index):
        u = np.ascontiguousarray(u, dtype=np.uint16)
        v = np.ascontiguousarray(v, dtype=np.uint16)
        s = np.ascontiguousarray(sample_index, dtype=np.uint32)
        out = np.zeros(u.size, dtype=PATH_DTYPE)
        self._check(self._lib.pt_trace_paths(self.handle, u.ctypes.data, v.ctypes.data, s.ctypes.data, u.size,
                                             out.ctypes.data))
        return out


def runtime_info(diag=False):
    """Which librccl / libamdhip64 the library's imports are bound to in THIS process, and their versions (pt_runtime_info)."""
    import json
    lib = load_library(diag)
    buf = C.create_string_buffer(2048)
    rc = lib.pt_runtime_info(buf, len(buf))
    if rc:
        raise PtError(rc, lib.pt_last_error(None).decode())
    return json.loads(buf.value.decode())


def comm_unique_id():
    """ncclGetUniqueId through the C-ABI: rank 0 makes it and hands the bytes to the other ranks."""
    lib = load_library()
    buf = (C.c_char * COMM_ID_BYTES)()
    rc = lib.pt_comm_get_unique_id(buf)
    if rc:
        raise PtError(rc, lib.pt_last_error(None).decode())
    return bytes(buf)


def worklist(width, height):
    """createWorkListForImage (src/LoadBalancer.cpp:38-52): one item per pixel, row-major (c, r)."""
    rec = np.zeros(width * height, dtype=TRACE_DTYPE)
    rr, cc = np.divmod(np.arange(width * height), width)
    rec["u"] = cc
    rec["v"] = rr
    return rec
