import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi
W,H=1104,1000
depth=int(sys.argv[1]) if len(sys.argv)>1 else 8
spp=64
r=ptmi.Renderer(W,H,max_path_length=depth,diag=(len(sys.argv)>2 and sys.argv[2]=='diag'))   # 'diag': the profiling build (honours PTMI_SERIAL etc.)
r.set_constant_env((1,1,1))
r.init_render_settings(samples_per_step=spp)
rec=ptmi.worklist(W,H); r.setup(rec)
r.path_trace()
t=time.time(); r.path_trace(); dt=time.time()-t
st=r.stats()
by=96.0*st.segments+88.0*st.escaped
print('const-env depth',depth,'sec %.4f'%dt,'Mpaths/s %.0f'%(st.paths/dt/1e6),'trace ms %.2f (sum of launches)'%st.path_trace_ms,'acc ms %.2f'%st.accumulate_ms,'alg GB/s %.0f'%(by/(st.path_trace_ms*1e-3)/1e9), 'seg/path %.2f'%(st.segments/st.paths))
