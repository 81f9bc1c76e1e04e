#!/bin/bash
# C4's full sample count (100k spp -> 334 steps of 300) AT C4'S WORLD SIZE through the drop-in CLI: EIGHT logical devices on ONE GPU
# (--devices 0 x 8: eight handles, eight host threads, resident film per device, image tiles traded between the eight by measured
# path length at every save interval, eight HDR tiles per gather through the host): a soak run of the multi-device host loop.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
A=/tmp/assets.extra; mkdir -p $A
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from ipu_path_trace_amd import nif_assets as N
N.write_metadata("$A/nif_metadata.txt"); N.write_ptnif("$A/converted.ptnif", N.synthetic_nif(), 12)
PY
make -C $ROOT/ipu_path_trace_amd/host -s
$ROOT/ipu_path_trace_amd/host/ipu_trace --assets $A -w 1104 -h 1000 -s 100000 --samples-per-step 300 --ipus 8 --devices 0,0,0,0,0,0,0,0 \
   --enable-load-balancing -o /tmp/image_c4c.png --save-interval 50 --max-path-length 8 > /tmp/c4c.log 2>&1
grep -c "Completed render step" /tmp/c4c.log
grep -c "Load balancing finished" /tmp/c4c.log
grep -E "share GPU|are padding|gathered through|Completed render step (1|100|200|334)/|Saved images at step (50|300)|Render finished|Samples/sec:" /tmp/c4c.log | cut -c1-190
