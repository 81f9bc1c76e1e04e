"""In-kernel clock of the fused NIF kernels on real data (MI355X_MICROARCH.md, "DVFS give-back" item 6): shader cycles
(s_memtime) over 100 MHz ticks (s_memrealtime) around workgroup 0's whole tile loop, read after >= 2 s of back-to-back
launches of the same kernel.  Loads libptmi_diag.so.  nif_kernel_v3 = the product kernel (32x32x16) built with its DIAG
bit 5; nif_kernel_v4 = its 16x16x32 twin (profiling build only).  usage: python scripts/clock_nif.py [seconds] [spp]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import nif_assets as A  # noqa: E402
from ipu_path_trace_amd import ptmi  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 120
W, H = 1104, 1000
lib = ptmi.load_library(diag=True)


def make(kernel):
    if kernel == "v4":
        os.environ["PTMI_NIF_KERNEL"] = "v4"
    r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
    r.init_nif_weights(A.synthetic_nif(), 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
    os.environ.pop("PTMI_NIF_KERNEL", None)
    r.init_render_settings(samples_per_step=spp)
    r.setup(ptmi.worklist(W, H))
    return r


for name, env in (("nif_kernel_v3 (32x32x16, product)", {"PTMI_NIF_DIAG": "32"}), ("nif_kernel_v4 (16x16x32)", {})):
    r = make("v4" if "v4" in name else "v3")
    os.environ.update(env)
    t0, steps, tf, ms = time.time(), 0, [], []
    while time.time() - t0 < seconds:
        r.path_trace()
        st = r.stats()
        tf.append(st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12)
        ms.append(st.total_ms)
        steps += 1
    out = (C.c_ulonglong * 2)()
    assert lib.pt_diag_nif_clock(r.handle, out) == 0
    for k in env:
        os.environ.pop(k, None)
    print("%-34s %d steps of %d spp in %.1f s | in-kernel clock %.3f GHz (%d cycles / %d ticks) | NIF %.1f TFLOP/s (last 3: %s) | step %.1f ms"
          % (name, steps, spp, time.time() - t0, out[0] / out[1] * 0.1, out[0], out[1], sum(tf[-3:]) / len(tf[-3:]),
             " ".join("%.0f" % x for x in tf[-3:]), sum(ms[-3:]) / len(ms[-3:])), flush=True)
    r.close()
