import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi, nif_assets as A
W,H,spp=1104,1000,300
r=ptmi.Renderer(W,H,max_path_length=8)
r.init_nif_weights(A.synthetic_nif(),12,A.URBAN_ALLEY_META['max'],A.folded_mean())
r.init_render_settings(samples_per_step=spp)
rec=ptmi.worklist(W,H); r.setup(rec); r.path_trace()
t=time.time()
for _ in range(3):
    r.setup(rec); r.path_trace(); r.read_results(rec)
dt=(time.time()-t)/3
t=time.time()
for _ in range(3): r.path_trace()
dt2=(time.time()-t)/3
print('per step incl. setup (H2D 22 MB) + read_results (D2H 22 MB): %.1f ms = %.1f Mpath-samples/s; resident: %.1f ms = %.1f M/s'%(dt*1e3, W*H*spp/dt/1e6, dt2*1e3, W*H*spp/dt2/1e6))
