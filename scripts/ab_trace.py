"""Same-process A/B of the trace kernel (profiling build): the round-3 kernel (PTMI_TRACE_KERNEL=opt0) and any variant named in
AB_TRACE_EXTRA (diag/ptmi_trace_variants.h: opt1 opt2 pipe scenec fn3 rounds cut1 cut2) against the product kernel, (a) on its own -- constant sky, so no NIF kernel runs beside it -- and (b) inside the C2 step (6x320 NIF).
usage: python scripts/ab_trace.py [rounds] [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import nif_assets as A  # noqa: E402
from ipu_path_trace_amd import ptmi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 300
W, H = 1104, 1000


def make(const_env):
    r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
    if const_env:
        r.set_constant_env((1.0, 1.0, 1.0))
    else:
        r.init_nif_weights(A.synthetic_nif(), 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    r.setup(ptmi.worklist(W, H))
    return r


for what, r in (("trace stage alone (constant sky)", make(True)), ("C2 step (NIF 6x320 beside it)", make(False))):
    res = {k: [] for k in ["opt0", "product"] + os.environ.get("AB_TRACE_EXTRA", "").split()}
    for rd in range(rounds + 1):
        for name in res:
            os.environ.pop("PTMI_TRACE_KERNEL", None)
            if name != "product":
                os.environ["PTMI_TRACE_KERNEL"] = name
            t = time.time()
            r.path_trace()
            dt = time.time() - t
            st = r.stats()
            if rd:
                res[name].append((st.path_trace_ms, dt * 1e3, st.paths / dt / 1e6))
    os.environ.pop("PTMI_TRACE_KERNEL", None)
    print(what)
    for name, v in res.items():
        print("  %-10s trace-kernel ms/step %s | step ms %s | Mpath/s %s" % (
            name, " ".join("%7.2f" % x[0] for x in v), " ".join("%7.2f" % x[1] for x in v), " ".join("%8.1f" % x[2] for x in v)), flush=True)
    r.close()
