"""One timed step of every BASELINE.json config shape on one GPU (synthetic NIF weights), for the record in profiles/.

C1 256x256, 16 spp, depth 4, constant sky      C2 1104x1000, 6x320 NIF, 300 spp, depth 8
C3 3840x2160, 6x320 NIF, 1000 spp, depth 16     C5 1104x1000, 8x1024 NIF, 300 spp, depth 8
(C4 is C2's image at 100k spp over 8 GPUs: see scripts/cli_c4_soak.sh for its sample count on one GPU.)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi, nif_assets as A

def run(name, W, H, spp, depth, nif=None, reps=2):
    r = ptmi.Renderer(W, H, max_path_length=depth)
    if nif is None:
        r.set_constant_env((0.5, 0.7, 1.0))
    else:
        r.init_nif_weights(nif, 12, A.URBAN_ALLEY_META["max"], A.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    rec = ptmi.worklist(W, H)
    r.setup(rec)
    r.path_trace()                       # warm-up step
    t = time.time()
    for _ in range(reps):
        r.path_trace()
    dt = (time.time() - t) / reps
    st = r.stats()
    tf = st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12 if st.nif_ms > 0 else 0.0
    print("%s  %dx%d  %d spp/step  depth %d:  %.1f ms/step  %.1f Mpath-samples/s  rays/s %.3g  NIF %.1f TFLOP/s  "
          "(escaped %.3f, segments/path %.2f; trace kernels %.1f ms, NIF %.1f ms, accumulate %.1f ms of HIP-event time; %s)"
          % (name, W, H, spp, depth, dt * 1e3, st.paths / dt / 1e6, st.segments / dt, tf, st.escaped / st.paths,
             st.segments / st.paths, st.path_trace_ms, st.nif_ms, st.accumulate_ms, r.nif_kernel_name() or "constant sky"), flush=True)
    r.close()

run("C1", 256, 256, 16, 4, None, reps=20)
run("C1 at C2 size", 1104, 1000, 300, 8, None)
run("C2", 1104, 1000, 300, 8, A.synthetic_nif())
run("C3", 3840, 2160, 1000, 16, A.synthetic_nif(), reps=1)
run("C3 constant sky (trace stage alone: the deep-path divergence stress)", 3840, 2160, 1000, 16, None, reps=1)
run("C2 image at depth 16, constant sky", 1104, 1000, 300, 16, None)
run("C5", 1104, 1000, 300, 8, A.synthetic_nif(hidden=1024, layer_count=8), reps=1)
