"""Interleaved A/B of NIF kernel variants of the profiling build, in ONE process on one device (cdna_hip_programming.md
section 5.4 rule 24).  usage (always loads libptmi_diag.so): python scripts/ab_nif.py [rounds] [spp] name=ENV:VAL ...
e.g.  v3= v2w8=PTMI_NIF_VARIANT:3 v2w4x64=PTMI_NIF_VARIANT:2 halfreads=PTMI_NIF_DIAG:64
Prints per variant the NIF TFLOP/s (escaped x FLOP / sum of NIF-kernel HIP-event time) and Mpath-samples/s of every round."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import nif_assets as A  # noqa: E402
from ipu_path_trace_amd import ptmi  # noqa: E402

args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 4
spp = int(args.pop(0)) if args and args[0].isdigit() else 150
variants = []
for a in args:
    name, _, env = a.partition("=")
    kv = dict(x.split(":") for x in env.split(",") if x)
    variants.append((name, kv))
hidden = int(os.environ.get("AB_HIDDEN", "320"))
layers = int(os.environ.get("AB_LAYERS", "6"))
W, H = 1104, 1000
r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
r.init_nif_weights(A.synthetic_nif(hidden=hidden, layer_count=layers), 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
r.init_render_settings(samples_per_step=spp)
rec = ptmi.worklist(W, H)
r.setup(rec)
res = {n: [] for n, _ in variants}
keys = sorted({k for _, kv in variants for k in kv})
for rd in range(rounds + 1):
    for name, kv in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(kv)
        t = time.time()
        r.path_trace()
        dt = time.time() - t
        st = r.stats()
        if rd:   # round 0 warms up
            res[name].append((st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12, st.paths / dt / 1e6, st.nif_ms, st.path_trace_ms))
for name, _ in variants:
    v = res[name]
    print("%-14s NIF TFLOP/s %s | Mpath/s %s | nif ms %s" % (
        name, " ".join("%7.1f" % x[0] for x in v), " ".join("%7.1f" % x[1] for x in v), " ".join("%6.1f" % x[2] for x in v)), flush=True)
