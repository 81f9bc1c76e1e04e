#!/bin/bash
# C4's full sample count (100k spp -> 334 steps of 300) through the drop-in CLI on ONE GPU: a soak run of the step loop
# (about 95 s), with the reference README's command shape.  Prints the final log lines and the output files.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
A=/tmp/assets.extra; mkdir -p $A
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from ipu_path_trace_amd import nif_assets as N
N.write_metadata("$A/nif_metadata.txt"); N.write_ptnif("$A/converted.ptnif", N.synthetic_nif(), 12)
PY
make -C $ROOT/ipu_path_trace_amd/host -s
$ROOT/ipu_path_trace_amd/host/ipu_trace --assets $A -w 1104 -h 1000 -s 100000 --samples-per-step 300 --ipus 1 --defer-attach \
   -o /tmp/image_c4.png --save-interval 50 --save-exe pt_graph --max-path-length 8 > /tmp/c4.log 2>&1
grep -c "Completed render step" /tmp/c4.log
grep -E "Completed render step (1|100|200|334)/|Saved images at step (50|300)|Render finished|Samples/sec:" /tmp/c4.log | cut -c1-170
ls -la /tmp/image_c4*
