import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi, nif_assets as A
W,H=1104,1000
spp=int(sys.argv[1]) if len(sys.argv)>1 else 8
r=ptmi.Renderer(W,H,max_path_length=8,diag='PTMI_GEMM_DIAG' in os.environ)   # a PTMI_GEMM_DIAG variant lives in the profiling build only
L=A.synthetic_nif(hidden=1024,layer_count=8)
r.init_nif_weights(L,12,A.URBAN_ALLEY_META['max'],A.folded_mean())
r.init_render_settings(samples_per_step=spp)
rec=ptmi.worklist(W,H); r.setup(rec)
r.path_trace()
t=time.time(); r.path_trace(); dt=time.time()-t
st=r.stats()
print('C5 8x1024 spp',spp,'sec %.3f'%dt,'Mpaths/s %.1f'%(st.paths/dt/1e6),'NIF TFLOP/s %.1f'%(st.escaped*st.nif_flops_per_sample/(st.nif_ms*1e-3)/1e12), 'nif ms %.1f'%st.nif_ms)
