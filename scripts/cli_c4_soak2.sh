#!/bin/bash
# C4's full sample count (100k spp -> 334 steps of 300) through the drop-in CLI with TWO logical devices on ONE GPU
# (--devices 0,0: the whole --ipus N step loop, tile trading by path length at every save interval, HDR tiles through the
# host): a soak run of the multi-device host loop.  Prints the final log lines.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
A=/tmp/assets.extra; mkdir -p $A
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from ipu_path_trace_amd import nif_assets as N
N.write_metadata("$A/nif_metadata.txt"); N.write_ptnif("$A/converted.ptnif", N.synthetic_nif(), 12)
PY
make -C $ROOT/ipu_path_trace_amd/host -s
$ROOT/ipu_path_trace_amd/host/ipu_trace --assets $A -w 1104 -h 1000 -s 100000 --samples-per-step 300 --ipus 2 --devices 0,0 \
   --enable-load-balancing -o /tmp/image_c4b.png --save-interval 50 --max-path-length 8 > /tmp/c4b.log 2>&1
grep -c "Completed render step" /tmp/c4b.log
grep -c "Load balancing finished" /tmp/c4b.log
grep -E "share GPU|Completed render step (1|100|200|334)/|Saved images at step (50|300)|Render finished|Samples/sec:" /tmp/c4b.log | cut -c1-170
