"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  The product package (ipu_path_trace_amd) never imports this module.
PARITY UNPINNED: see oracle/pt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")
_FAST_PATH = os.path.join(_HERE, "libpt_oracle_fast.so")
_FAST_STAMP = os.path.join(_HERE, "libpt_oracle_fast.stamp")


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _host_stamp():
    """What -march=native depends on: this host's CPU flags (and the source's age)."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    src = os.path.join(_HERE, "pt_oracle.c")
    return hashlib.sha256((flags + str(os.path.getmtime(src))).encode()).hexdigest()


def build_fast(force=False):
    """The TIMING build of the same source (oracle/Makefile: -O3 -march=native -fopenmp, -DORC_FAST_BUILD) -- only for the
    cpu_baseline leg of bench.py and its own CPU test; never the parity checker.  -march=native is host-specific, so the
    library is rebuilt whenever it was made for a host with other CPU flags (a .so built in the container travels to the
    GPU box, whose CPU is a different one)."""
    stamp = _host_stamp()
    old = None
    if os.path.exists(_FAST_STAMP):
        with open(_FAST_STAMP) as f:
            old = f.read().strip()
    if force or not os.path.exists(_FAST_PATH) or old != stamp:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpt_oracle_fast.so"], stdout=subprocess.DEVNULL)
        with open(_FAST_STAMP, "w") as f:
            f.write(stamp)
    return _FAST_PATH


class TraceRecord(C.Structure):
    """src/codelets/TraceRecord.hpp:7-19"""
    _fields_ = [("u", C.c_uint16), ("v", C.c_uint16), ("r", C.c_float), ("g", C.c_float), ("b", C.c_float),
                ("sampleCount", C.c_uint16), ("pathLength", C.c_uint16)]


TRACE_DTYPE = np.dtype([("u", "<u2"), ("v", "<u2"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"),
                        ("sampleCount", "<u2"), ("pathLength", "<u2")], align=True)
assert TRACE_DTYPE.itemsize == 20 and C.sizeof(TraceRecord) == 20


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_path_length", C.c_uint32),
                ("roulette_depth", C.c_uint32), ("stop_prob", C.c_float), ("refractive_index", C.c_float),
                ("aa_noise_type", C.c_int32), ("sample_precision", C.c_int32), ("seed", C.c_uint64),
                ("aa_noise_scale", C.c_float), ("fov_radians", C.c_float), ("azimuth_radians", C.c_float),
                ("env_mode", C.c_int32), ("env_rgb", C.c_float * 3), ("fold", C.c_int32)]


class Layer(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("kernel", C.c_void_p), ("bias", C.c_void_p),
                ("relu", C.c_int32)]


class LayerAny(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("kernel", C.c_void_p), ("bias", C.c_void_p),
                ("relu", C.c_int32), ("float32", C.c_int32)]


class Path(C.Structure):
    _fields_ = [("length", C.c_uint32), ("escaped", C.c_uint32), ("dir", C.c_float * 3), ("uv", C.c_float * 2),
                ("throughput", C.c_float * 3), ("cam", C.c_float * 2)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("escaped", C.c_uint64)]


AA_NORMAL, AA_UNIFORM, AA_TRUNCATED_NORMAL = 0, 1, 2
SAMPLES_HALF, SAMPLES_FLOAT = 0, 1
ENV_CONSTANT, ENV_NIF = 0, 1
FOLD_BACKWARD, FOLD_FORWARD = 0, 1
DIFFUSE, EMIT, ESCAPED, REFRACT, SPECULAR, DEBUG, END, SKIP = range(8)

_lib = None
_fast_lib = None


def lib(fast=False):
    """fast=False: the strict build, the parity checker.  fast=True: the timing build (build_fast)."""
    global _lib, _fast_lib
    if fast:
        if _fast_lib is None:
            _fast_lib = _bind(C.CDLL(build_fast()))
        return _fast_lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    if True:
        fp = C.POINTER(C.c_float)
        L.orc_nif_create.restype = C.c_void_p
        L.orc_nif_create.argtypes = [C.POINTER(Layer), C.c_uint32, C.c_uint32, C.c_float, fp, C.c_int32]
        L.orc_nif_create_f32.restype = C.c_void_p
        L.orc_nif_create_f32.argtypes = [C.POINTER(Layer), C.c_uint32, C.c_uint32, C.c_float, fp, C.c_int32]
        L.orc_nif_create_mixed.restype = C.c_void_p
        L.orc_nif_create_mixed.argtypes = [C.POINTER(LayerAny), C.c_uint32, C.c_uint32, C.c_float, fp, C.c_int32]
        L.orc_nif_destroy.argtypes = [C.c_void_p]
        L.orc_nif_infer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_nif_encode.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_nif_flops_per_sample.restype = C.c_uint64
        L.orc_nif_flops_per_sample.argtypes = [C.c_void_p]
        L.orc_trace_path.argtypes = [C.POINTER(Config), C.c_uint16, C.c_uint16, C.c_uint32, C.POINTER(Path)]
        L.orc_trace_records.argtypes = [C.POINTER(Config), C.c_uint16, C.c_uint16, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_render.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                 C.POINTER(Stats)]
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_f2h.restype = C.c_uint16
        L.orc_f2h.argtypes = [C.c_float]
        L.orc_h2f.restype = C.c_float
        L.orc_h2f.argtypes = [C.c_uint16]
        L.orc_dm_log.restype = C.c_float
        L.orc_dm_log.argtypes = [C.c_float]
        L.orc_dm_sincos2pi.argtypes = [C.c_float, fp, fp]
        L.orc_dm_atan2.restype = C.c_float
        L.orc_dm_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_dm_acos.restype = C.c_float
        L.orc_dm_acos.argtypes = [C.c_float]
        L.orc_pixel_to_ray.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_intersect_sphere.restype = C.c_float
        L.orc_intersect_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.orc_intersect_disc.restype = C.c_float
        L.orc_intersect_disc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.orc_scene_intersect.argtypes = [C.c_void_p, C.c_void_p, fp, C.c_void_p, C.c_void_p]
        L.orc_reflect.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.orc_hemisphere.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.orc_diffuse_dir.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_roulette.argtypes = [C.c_float, C.c_float, fp]
        L.orc_dir_to_uv.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.orc_aa_noise.argtypes = [C.POINTER(Config), C.c_uint16, C.c_uint16, C.c_uint32, C.c_void_p]
        L.orc_scene_object.argtypes = [C.c_int, C.c_void_p, fp, C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_object_ids.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_specular_ids.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_build_info.restype = C.c_char_p
        L.orc_max_threads.restype = C.c_int
    return L


def make_config(width=256, height=256, max_path_length=10, roulette_depth=3, stop_prob=0.3,
                refractive_index=1.5, aa_noise_type=AA_NORMAL, sample_precision=SAMPLES_HALF, seed=1,
                aa_noise_scale=0.3, fov_degrees=90.0, env_rotation_degrees=0.0, env_mode=ENV_CONSTANT,
                env_rgb=(1.0, 1.0, 1.0), fold=FOLD_FORWARD):
    """Defaults are the reference CLI defaults (PathTracerApp.cpp:797-817); degree->radian
    conversions follow PathTracerApp.cpp:574,584 in single precision."""
    cfg = Config()
    cfg.width, cfg.height = width, height
    cfg.max_path_length, cfg.roulette_depth = max_path_length, roulette_depth
    cfg.stop_prob, cfg.refractive_index = stop_prob, refractive_index
    cfg.aa_noise_type, cfg.sample_precision, cfg.seed = aa_noise_type, sample_precision, seed
    cfg.aa_noise_scale = aa_noise_scale
    cfg.fov_radians = float(np.float32(fov_degrees) * np.float32(np.pi / 180.0))
    cfg.azimuth_radians = float(np.float32((np.float32(env_rotation_degrees) / np.float32(360.0)) * (2.0 * np.pi)))
    cfg.env_mode = env_mode
    cfg.env_rgb[:] = env_rgb
    cfg.fold = fold
    return cfg


class Nif:
    """Owns an orc_nif built from (kernel [in,out], bias [out] | None, relu) triples.  Every layer runs in the type of its own
    kernel (the reference gives a matmul its kernel's type, NifModel.cpp:314): a float32 kernel -> that layer in float, anything
    else -> binary16, with the activations cast between layers of different types (orc_nif_create_mixed)."""

    def __init__(self, layers, embedding_dim, max_value, mean_folded, log_tonemap=True, fast=False):
        self._keep = []
        self._lib = lib(fast)
        arr = (LayerAny * len(layers))()
        kinds = [np.asarray(k).dtype == np.float32 for k, _, _ in layers]
        self.float32 = all(kinds)
        self.mixed = any(kinds) and not all(kinds)
        for i, (k, b, relu) in enumerate(layers):
            dt = np.float32 if kinds[i] else np.float16
            k = np.ascontiguousarray(k, dtype=dt)
            self._keep.append(k)
            arr[i].rows, arr[i].cols = k.shape
            arr[i].kernel = k.ctypes.data
            if b is not None:
                b = np.ascontiguousarray(b, dtype=dt)
                self._keep.append(b)
                arr[i].bias = b.ctypes.data
            arr[i].relu = int(bool(relu))
            arr[i].float32 = int(kinds[i])
        mean = (C.c_float * 3)(*[float(x) for x in mean_folded])
        self.handle = self._lib.orc_nif_create_mixed(arr, len(layers), embedding_dim, float(max_value), mean, int(log_tonemap))
        self.embedding_dim = embedding_dim

    def __del__(self):
        if getattr(self, "handle", None):
            self._lib.orc_nif_destroy(self.handle)
            self.handle = None

    def infer(self, u, v):
        u = np.ascontiguousarray(u, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.empty((u.size, 3), dtype=np.float32)
        rc = self._lib.orc_nif_infer(self.handle, u.ctypes.data, v.ctypes.data, u.size, out.ctypes.data)
        if rc:
            raise RuntimeError("orc_nif_infer failed: %d" % rc)
        return out

    def flops_per_sample(self):
        return int(self._lib.orc_nif_flops_per_sample(self.handle))


def nif_encode(embedding_dim, u, v):
    out = np.empty(4 * embedding_dim, dtype=np.float32)
    lib().orc_nif_encode(embedding_dim, float(u), float(v), out.ctypes.data)
    return out


def trace_path(cfg, u, v, sample):
    p = Path()
    lib().orc_trace_path(C.byref(cfg), u, v, sample, C.byref(p))
    return p


def trace_records(cfg, u, v, sample, capacity=64):
    types = np.zeros(capacity, dtype=np.int32)
    clr = np.zeros((capacity, 3), dtype=np.float32)
    w = np.zeros(capacity, dtype=np.float32)
    n = lib().orc_trace_records(C.byref(cfg), u, v, sample, types.ctypes.data, clr.ctypes.data, w.ctypes.data, capacity)
    return types[:n], clr[:n], w[:n]


def render(cfg, nif, records, sample_base, n_samples, fast=False):
    """records: numpy array of TRACE_DTYPE, updated in place.  Returns Stats.  fast=True: the timing build (the NIF, if any,
    must have been created with fast=True as well)."""
    assert records.dtype == TRACE_DTYPE and records.flags.c_contiguous
    assert nif is None or nif._lib is lib(fast), "a Nif belongs to the build that created it"
    st = Stats()
    rc = lib(fast).orc_render(C.byref(cfg), nif.handle if nif is not None else None, records.ctypes.data, records.size,
                          sample_base, n_samples, C.byref(st))
    if rc:
        raise RuntimeError("orc_render failed: %d" % rc)
    return st


def object_ids(width, height, fov_radians):
    """int8 [height, width]: index of the object the central ray of each pixel hits (-1 = environment)."""
    out = np.empty((height, width), dtype=np.int8)
    lib().orc_object_ids(width, height, float(fov_radians), out.ctypes.data)
    return out


def specular_ids(width, height, fov_radians, refractive_index=1.5, reflect_variant=0, max_bounces=8):
    """(int8 [height, width] final object, uint8 [height, width] specular interactions): the central ray of each pixel followed
    deterministically through mirror (reflect) and glass (refract) to the first diffuse object (0..5), the environment (-1) or the
    bounce limit (-2).  reflect_variant 1 and refractive indices other than the scene's 1.5 are the tests' negative controls."""
    ids = np.empty((height, width), dtype=np.int8)
    nb = np.empty((height, width), dtype=np.uint8)
    lib().orc_specular_ids(width, height, float(fov_radians), float(refractive_index), int(reflect_variant), int(max_bounces),
                           ids.ctypes.data, nb.ctypes.data)
    return ids, nb


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def _v3(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def worklist(width, height):
    """LoadBalancer.cpp:38-52 createWorkListForImage: row-major (c, r)."""
    rec = np.zeros(width * height, dtype=TRACE_DTYPE)
    rr, cc = np.divmod(np.arange(width * height), width)
    rec["u"] = cc
    rec["v"] = rr
    return rec
