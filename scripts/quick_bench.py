import numpy as np, time, sys
sys.path.insert(0,'/root/repo')
from ipu_path_trace_amd import ptmi, nif_assets as A
W,H=1104,1000
r=ptmi.Renderer(W,H,max_path_length=8)
L=A.synthetic_nif()
r.init_nif_weights(L,12,A.URBAN_ALLEY_META['max'],A.folded_mean())
for spp in (8,32,300):
    r.init_render_settings(samples_per_step=spp)
    rec=ptmi.worklist(W,H); r.setup(rec)
    t=time.time(); r.path_trace(); dt=time.time()-t
    st=r.stats()
    print(spp, 'sec',dt,'Mpaths/s',st.paths/dt/1e6, st.as_dict(), flush=True)
    print('  NIF TFLOP/s', st.escaped*st.nif_flops_per_sample/(st.nif_ms*1e-3)/1e12, 'trace ms',st.path_trace_ms,'nif ms',st.nif_ms,'acc ms',st.accumulate_ms)
