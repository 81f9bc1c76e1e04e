"""The accumulate pass on its own: constant sky, so the step is trace + accumulate only.
  serial  (profiling build, PTMI_SERIAL=1: every kernel on one stream) -> accumulate_ms is the kernel ALONE; bytes per launch =
          13 B per path (1 B path record + 12 B radiance) + 40 B per work item (five 4-byte accumulators read and written)
  product (libptmi.so, three streams) -> the constant-sky step as the product runs it
usage: python scripts/acc_bench.py [c2|c3] [spp] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H, depth, spp = (3840, 2160, 16, 1000) if cfg == "c3" else (1104, 1000, 8, 300)
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for mode in ("serial", "product"):
    if mode == "serial":
        os.environ["PTMI_SERIAL"] = "1"
    r = ptmi.Renderer(W, H, max_path_length=depth, diag=(mode == "serial"))
    os.environ.pop("PTMI_SERIAL", None)
    r.set_constant_env((1.0, 1.0, 1.0))
    r.init_render_settings(samples_per_step=spp)
    r.setup(ptmi.worklist(W, H))
    r.path_trace()
    for _ in range(steps):
        t = time.perf_counter()
        r.path_trace()
        dt = time.perf_counter() - t
        st = r.stats()
        n = W * H
        byts = st.paths * 13 + st.accumulate_launches * n * 40
        print("%s %-7s %dx%d %d spp depth %d: step %.2f ms (device %.2f) | trace %.2f ms in %d launches | accumulate %.3f ms in %d launches = %.0f GB/s of %.2f GB algorithmic"
              % (cfg, mode, W, H, spp, depth, dt * 1e3, st.total_ms, st.path_trace_ms, st.trace_launches, st.accumulate_ms, st.accumulate_launches,
                 byts / (st.accumulate_ms * 1e-3) / 1e9, byts / 1e9), flush=True)
    r.close()
