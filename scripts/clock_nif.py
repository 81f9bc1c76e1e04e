"""In-kernel clock of the fused NIF kernel on real data (MI355X_MICROARCH.md, "DVFS give-back" item 6): shader cycles
(s_memtime) over 100 MHz ticks (s_memrealtime) around workgroup 0's whole tile loop, read after >= 2 s of back-to-back
launches of the same kernel.  Loads libptmi_diag.so.  Variants are values of PTMI_NIF_DIAG whose build carries the stamp
bit (32): 32 = the product kernel, 96 = half the LDS reads of A (the NB = 2 bound, timing only), 34 = no LDS reads of A
(timing only).  usage: python scripts/clock_nif.py [seconds] [spp] [diag values ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import nif_assets as A  # noqa: E402
from ipu_path_trace_amd import ptmi  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 120
diags = [int(x) for x in sys.argv[3:]] or [32, 96, 34]
NAMES = {32: "nif_kernel_v3 (product)", 96: "half the LDS reads of A (NB = 2 bound)", 34: "no LDS reads of A"}
W, H = 1104, 1000
lib = ptmi.load_library(diag=True)
r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
r.init_nif_weights(A.synthetic_nif(), 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
r.init_render_settings(samples_per_step=spp)
r.setup(ptmi.worklist(W, H))
for d in diags:
    assert d & 32, "the variant must carry the clock-stamp bit (32)"
    os.environ["PTMI_NIF_DIAG"] = str(d)
    t0, steps, tf, ms = time.time(), 0, [], []
    while time.time() - t0 < seconds:
        r.path_trace()
        st = r.stats()
        tf.append(st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12)
        ms.append(st.total_ms)
        steps += 1
    out = (C.c_ulonglong * 2)()
    assert lib.pt_diag_nif_clock(r.handle, out) == 0
    os.environ.pop("PTMI_NIF_DIAG", None)
    print("%-42s %d steps of %d spp in %.1f s | in-kernel clock %.3f GHz (%d cycles / %d ticks) | NIF %.1f TFLOP/s (last 3: %s) | step %.1f ms"
          % (NAMES.get(d, "PTMI_NIF_DIAG=%d" % d), steps, spp, time.time() - t0, out[0] / out[1] * 0.1, out[0], out[1], sum(tf[-3:]) / len(tf[-3:]),
             " ".join("%.0f" % x for x in tf[-3:]), sum(ms[-3:]) / len(ms[-3:])), flush=True)
r.close()
