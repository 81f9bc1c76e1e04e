"""Same-process A/B of the wide-NIF layer path (BASELINE config C5: NIF 8 x 1024): the product's 16x16x32 kernels with the
fused head (pt_nif_gemm.h) against variants of themselves (PTMI_GEMM_DIAG bits, PTMI_CHUNK_STREAMS, CHUNK sizes), interleaved
round by round on one device (cdna_hip_programming.md section 5.4 rule 24).  The round-2 32x32x16 baseline kernels were
removed in round 5 (their numbers: profiles/r03_c5_ablation.txt).
Loads libptmi_diag.so.  usage: python scripts/ab_c5.py [rounds] [spp] [name=ENV:VAL,...]   (variants of the 16 path)
Prints per variant the NIF TFLOP/s (escaped x FLOP / sum of NIF-stage HIP-event time) and Mpath-samples/s of every round."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import nif_assets as A  # noqa: E402
from ipu_path_trace_amd import ptmi  # noqa: E402

args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 4
spp = int(args.pop(0)) if args and args[0].isdigit() else 30
extra = []
for a in args:
    name, _, env = a.partition("=")
    extra.append((name, dict(x.split(":") for x in env.split(",") if x)))
hidden = int(os.environ.get("AB_HIDDEN", "1024"))
layers = int(os.environ.get("AB_LAYERS", "8"))
W, H = 1104, 1000
L = A.synthetic_nif(hidden=hidden, layer_count=layers)


def make(chunk=None):
    if chunk:
        os.environ["PTMI_GEMM_CHUNK"] = str(chunk)
    r = ptmi.Renderer(W, H, max_path_length=8, diag=True)
    r.init_nif_weights(L, 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
    os.environ.pop("PTMI_GEMM_CHUNK", None)
    r.init_render_settings(samples_per_step=spp)
    r.setup(ptmi.worklist(W, H))
    return r


variants = [("mfma16x16x32", make(), {})]
for n, kv in extra:   # CHUNK:<tiles> in a variant makes its own renderer (the chunk size is fixed when the weights are uploaded)
    chunk = kv.pop("CHUNK", None)
    variants.append((n, make(int(chunk)) if chunk else variants[0][1], kv))
keys = sorted({k for _, _, kv in variants for k in kv})
res = {n: [] for n, _, _ in variants}
for rd in range(rounds + 1):
    for name, r, kv in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(kv)
        t = time.time()
        r.path_trace()
        dt = time.time() - t
        st = r.stats()
        if rd:   # round 0 warms up
            res[name].append((st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12, st.paths / dt / 1e6, st.nif_ms))
            if int(kv.get("PTMI_GEMM_DIAG", "0")) & 32:   # in-kernel clock of the last layer launch: shader cycles per 100 MHz tick
                import ctypes as C
                out = (C.c_ulonglong * 256)()
                assert ptmi.load_library(diag=True).pt_diag_stamps(r.handle, out) == 0
                if out[1]:
                    print("  %s: in-kernel clock %.3f GHz (%d shader cycles in %d ticks of 10 ns, workgroup 0 of the last layer launch)"
                          % (name, out[0] / out[1] * 0.1, out[0], out[1]), flush=True)
for name, _, _ in variants:
    v = res[name]
    print("%-14s NIF TFLOP/s %s | Mpath/s %s | nif ms %s" % (
        name, " ".join("%7.1f" % x[0] for x in v), " ".join("%6.2f" % x[1] for x in v), " ".join("%7.1f" % x[2] for x in v)), flush=True)
