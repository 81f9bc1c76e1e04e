import sys, time, faulthandler, os
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.modules["torch"] = None
from ipu_path_trace_amd import ptmi
print("RUNTIME", ptmi.runtime_info(), flush=True)
r = ptmi.Renderer(32, 32, max_path_length=4)
print("renderer ok", flush=True)
r.comm_set_timeout(4000)
uid = ptmi.comm_unique_id()
print("uid ok", flush=True)
t = time.time()
try:
    r.comm_init_rank(uid, 0, 2)
    print("UNEXPECTED")
except ptmi.PtError as e:
    print("CODE", e.code, "AFTER %.1f" % (time.time() - t), "MSG", e, flush=True)
r.set_constant_env((1.0, 1.0, 1.0))
r.init_render_settings(samples_per_step=2)
rec = ptmi.worklist(32, 32)
r.setup(rec)
r.path_trace()
print("path_trace ok", flush=True)
try:
    r.gather_hdr(32 * 32)
except ptmi.PtError as e:
    print("gather refused as expected", e, flush=True)
r.comm_init_rank(ptmi.comm_unique_id(), 0, 1)
print("world-1 comm ok", flush=True)
tiles = r.gather_hdr(32 * 32)
print("gather ok", tiles.shape, flush=True)
r.close()
print("LONE_RANK_OK", flush=True)
