#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_trace.sh <tag> [depth = 8]
# Counter passes (counters only, no tracing domains) over the trace stage on its own: scripts/trace_bench.py renders two
# 64-spp constant-sky steps of the 1104x1000 worklist at depth 8, so trace_kernel runs with no NIF kernel beside it.
# scripts/summarize_trace_pmc.py <tag> turns the CSVs into profiles/<tag>_trace_pmc.json (read by bench.py).
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DEPTH=${2:-8}
CMD="python3 $ROOT/scripts/trace_bench.py $DEPTH"
echo $DEPTH > $OUT/depth
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o c -- $CMD > $OUT/sq.log 2>&1
echo "[pmc_trace] SQ pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- $CMD > $OUT/fetch.log 2>&1
echo "[pmc_trace] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- $CMD > $OUT/write.log 2>&1
echo "[pmc_trace] WRITE_SIZE done"
