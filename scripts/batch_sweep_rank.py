"""Batch-size sweep at the per-rank size of an 8-GPU run: 1/8 of the C2 image through the tile partition."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi, partition, nif_assets as A
W, H, spp = 1104, 1000, 300
L = A.synthetic_nif()
work = partition.tile_order_worklist(W, H, 3, 8)
for k in [int(x) for x in sys.argv[1:]]:
    r = ptmi.Renderer(W, H, max_work_items=work.size, max_path_length=8, iterations_per_batch=k)
    r.init_nif_weights(L, 12, A.URBAN_ALLEY_META['max'], A.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    r.setup(work)
    r.path_trace(); r.path_trace()
    t = time.time()
    for _ in range(8): r.path_trace()
    dt = (time.time() - t) / 8
    st = r.stats()
    print('rank share %d px, iterations_per_batch %d: ms/step %.2f  Mpaths/s x8 %.1f  NIF TFLOP/s %.1f  launches %d' % (
        work.size, k, dt * 1e3, 8 * st.paths / dt / 1e6, st.escaped * st.nif_flops_per_sample / (st.nif_ms * 1e-3) / 1e12, st.nif_launches), flush=True)
    r.close()
