"""Profiling build: the trace kernel's grid size (workgroups = queue regions; product: up to 2048 = 8 per CU) against the stand-alone
trace time and the C2 step.  usage: python scripts/sweep_trace_blocks.py [blocks ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import nif_assets as A, ptmi
W, H = 1104, 1000
for blocks in [int(x) for x in sys.argv[1:]] or [0, 2048, 1536, 1280, 1024, 768, 512]:   # 0 = the library's own choice (occupancy x CUs)
    os.environ.pop("PTMI_TRACE_BLOCKS", None)
    if blocks:
        os.environ["PTMI_TRACE_BLOCKS"] = str(blocks)
    out = []
    for const in (True, False):
        r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
        if const:
            r.set_constant_env((1.0, 1.0, 1.0))
        else:
            r.init_nif_weights(A.synthetic_nif(), 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
        r.init_render_settings(samples_per_step=300)
        r.setup(ptmi.worklist(W, H))
        r.path_trace()
        ts = []
        for _ in range(3):
            t = time.time(); r.path_trace(); dt = time.time() - t
            st = r.stats(); ts.append((st.path_trace_ms, dt * 1e3))
        out.append(min(ts))
        r.close()
    print("blocks %5d: alone trace %.2f ms (step %.2f) | C2 step: trace %.1f ms, step %.2f ms" % (blocks, out[0][0], out[0][1], out[1][0], out[1][1]), flush=True)
