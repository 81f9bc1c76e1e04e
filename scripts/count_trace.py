import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipu_path_trace_amd import ptmi
W,H=1104,1000
for depth in (8,16):
    r=ptmi.Renderer(W,H,max_path_length=depth,diag=True)
    r.set_constant_env((1,1,1)); r.init_render_settings(samples_per_step=64); r.setup(ptmi.worklist(W,H))
    os.environ["PTMI_TRACE_KERNEL"]="count"
    r.path_trace()
    st=r.stats()
    out=(C.c_ulonglong*256)()
    assert ptmi.load_library(diag=True).pt_diag_stamps(r.handle,out)==0
    trips,act,ttrips,tact=out[0],out[1],out[2],out[3]
    print("depth",depth,"paths",st.paths,"segments",st.segments,"wave-trips",trips,"active lane-trips",act,"occupancy %.3f"%(act/(64.0*trips)),
          "| tail (state list dry): wave-trips",ttrips,"(%.1f %% of trips)"%(100.0*ttrips/trips),"occupancy %.3f"%(tact/(64.0*max(ttrips,1))))
    r.close()
