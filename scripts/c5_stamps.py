"""Phase timing of the wide-NIF layer kernel from in-kernel s_memtime stamps (profiling build, PTMI_GEMM_DIAG=64).
usage (always loads libptmi_diag.so): python scripts/c5_stamps.py
Prints, for waves 0 and 4 of workgroup 0 over 16 stages of a hidden layer's second block: cycles from leaving a barrier to
reaching the next (the phase's own work) and cycles spent waiting in each barrier."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTMI_GEMM_DIAG", "64")   # 64: four phases per stage (product structure); 192: two phases
NPH = 2 if os.environ["PTMI_GEMM_DIAG"] == "192" else 4
from ipu_path_trace_amd import nif_assets as A, ptmi  # noqa: E402

W, H = 1104, 1000
r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
r.init_nif_weights(A.synthetic_nif(hidden=1024, layer_count=8), 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
r.init_render_settings(samples_per_step=8)
rec = ptmi.worklist(W, H)
r.setup(rec)
r.path_trace()
r.path_trace()
out = (C.c_ulonglong * 256)()
lib = ptmi.load_library(diag=True)
assert lib.pt_diag_stamps(r.handle, out) == 0
t = np.array(out, dtype=np.uint64).reshape(2, 16, 8).astype(np.int64)
names = ["MFMA k0", "load A", "MFMA k1", "load B"] if NPH == 4 else ["MFMA x16", "load"]
for w in range(2):
    work = np.zeros((15, NPH)); wait = np.zeros((15, NPH))
    for st in range(15):
        for ph in range(NPH):
            enter, leave = t[w, st, 2 * ph], t[w, st, 2 * ph + 1]
            prev_leave = t[w, st, 2 * ph - 1] if ph else t[w, st - 1, 2 * NPH - 1] if st else enter
            work[st, ph] = enter - prev_leave
            wait[st, ph] = leave - enter
    print("wave %d: cycles per phase (median over stages 1..14)" % (4 * w))
    for ph in range(NPH):
        print("   %-8s own work %6.0f   barrier wait %6.0f" % (names[ph], np.median(work[1:, ph]), np.median(wait[1:, ph])))
    print("   stage total %6.0f cycles (16 MFMAs = 512 pipe cycles per wave, 1024 per SIMD)" % np.median(t[w, 1:15, 2 * NPH - 1] - t[w, 0:14, 2 * NPH - 1]))
