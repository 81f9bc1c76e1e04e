import numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi, nif_assets as A
W,H=1104,1000
spps = [int(x) for x in sys.argv[1:]] or [32, 300]
DIAG = os.environ.get('QB_DIAG') == '1'   # the profiling build (libptmi_diag.so): the PTMI_* A/B switches only exist there
r=ptmi.Renderer(W,H,max_path_length=8,diag=DIAG)
L=A.synthetic_nif(hidden=int(os.environ.get('AB_HIDDEN','320')),layer_count=int(os.environ.get('AB_LAYERS','6')),
                dtype=np.float32 if os.environ.get('AB_DTYPE') == 'f32' else np.float16)
r.init_nif_weights(L,12,A.URBAN_ALLEY_META['max'],A.folded_mean())
for spp in spps:
    r.init_render_settings(samples_per_step=spp)
    rec=ptmi.worklist(W,H); r.setup(rec)
    r.path_trace()
    t=time.time(); r.path_trace(); dt=time.time()-t
    st=r.stats()
    print('variant',os.environ.get('PTMI_NIF_VARIANT','default'),'spp',spp,'sec %.4f'%dt,'Mpaths/s %.1f'%(st.paths/dt/1e6),
          'NIF TFLOP/s %.1f'%(st.escaped*st.nif_flops_per_sample/(st.nif_ms*1e-3)/1e12),
          'trace ms %.2f nif ms %.2f acc ms %.2f'%(st.path_trace_ms,st.nif_ms,st.accumulate_ms), flush=True)
