#!/bin/bash
# usage: scripts/pmc_c5.sh <outdir-name> <counters...>   (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass (counters only, no tracing domains) over a short C5 (NIF 8x1024) run.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/bench_c5.py 4 > $OUT.log 2>&1
