#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/profile_round.sh <tag>      e.g. r01_h
# Produces under gpurun_out/<tag>/: bench.json (plain run), kernel_stats.csv (rocprofv3 --kernel-trace --stats of the
# same bench command) and one counter CSV per --pmc pass (counters only, never combined with tracing domains).
# scripts/summarize_profile.py turns them into the files committed under profiles/.
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
echo "[profile_round] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace*" -delete
echo "[profile_round] kernel trace done"
SHORT="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- $SHORT > $OUT/pmc_fetch.log 2>&1
echo "[profile_round] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- $SHORT > $OUT/pmc_write.log 2>&1
echo "[profile_round] WRITE_SIZE done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -o c -- $SHORT > $OUT/pmc_mfma.log 2>&1
echo "[profile_round] MFMA counters done"
python3 $ROOT/scripts/summarize_profile.py $TAG $OUT/summary > $OUT/summary.log 2>&1 && echo "[profile_round] summarised into $OUT/summary"
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma   # raw per-dispatch counter CSVs: far beyond what gpurun copies back
find $OUT/trace -type f ! -name "*kernel_stats.csv" -delete
