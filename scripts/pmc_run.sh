#!/bin/bash
# usage: scripts/pmc_run.sh <outdir-name> <counters...>   (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass (counters only, no tracing domains) over a short bench run.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/quick_bench.py 16 > $OUT.log 2>&1
