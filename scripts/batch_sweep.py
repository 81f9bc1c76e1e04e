import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi, nif_assets as A
W,H,spp=1104,1000,300
L=A.synthetic_nif()
for k in [int(x) for x in sys.argv[1:]]:
    r=ptmi.Renderer(W,H,max_path_length=8,iterations_per_batch=k)
    r.init_nif_weights(L,12,A.URBAN_ALLEY_META['max'],A.folded_mean())
    r.init_render_settings(samples_per_step=spp)
    rec=ptmi.worklist(W,H); r.setup(rec)
    r.path_trace()
    t=time.time(); r.path_trace(); r.path_trace(); dt=(time.time()-t)/2
    st=r.stats()
    print('iterations_per_batch',k,'ms/step %.1f'%(dt*1e3),'Mpaths/s %.1f'%(st.paths/dt/1e6),'NIF TFLOP/s %.1f'%(st.escaped*st.nif_flops_per_sample/(st.nif_ms*1e-3)/1e12),'launches',st.nif_launches, flush=True)
    r.close()
