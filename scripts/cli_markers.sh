#!/bin/bash
# usage (GPU box, via gpurun): scripts/cli_markers.sh <tag>
# The drop-in CLI at the headline configuration under rocprofv3 --marker-trace --kernel-trace --stats: the host-phase
# ranges (host/trace_ranges.hpp: the reference's pvti::Tracepoint names) beside the kernels they enclose.  Tracing domains
# only, no counters.  The profiled program is ipu_trace itself (no shell, env or launcher between rocprofv3 and it).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
A=/tmp/assets.extra; mkdir -p $A
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from ipu_path_trace_amd import nif_assets as N
N.write_metadata("$A/nif_metadata.txt"); N.write_ptnif("$A/converted.ptnif", N.synthetic_nif(), 12)
PY
make -C $ROOT/ipu_path_trace_amd/host -s
cd /tmp && export TMPDIR=/tmp
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/markers -o m -- \
  $ROOT/ipu_path_trace_amd/host/ipu_trace --assets $A -w 1104 -h 1000 -s 1800 --samples-per-step 300 --ipus 1 \
  -o /tmp/image.png --save-interval 3 --max-path-length 8 --log-level debug > $OUT/cli.log 2>&1
find $OUT/markers -name "*kernel_trace*" -delete
find $OUT/markers -name "*marker_api_trace.csv" -size +2M -delete
find $OUT/markers -name "*agent_info*" -delete
ls $OUT/markers
grep -i "trace ranges\|Samples/sec" $OUT/cli.log | tail -4
for f in $(find $OUT/markers -name "*marker*stats*.csv" -o -name "*domain_stats.csv"); do echo "== $f"; head -30 $f; done
