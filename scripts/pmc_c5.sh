#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_c5.sh <tag>
# C5 (NIF 8x1024) profile set: kernel-trace stats of one short run plus separate counter passes (counters only, never
# combined with tracing domains) -- FETCH_SIZE, WRITE_SIZE and the MFMA-pipe counters -- over scripts/bench_c5.py 8.
# scripts/summarize_c5.py <tag> turns them into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json.
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/bench_c5.py 8"
$CMD > $OUT/plain.log 2>&1
echo "[pmc_c5] plain: $(tail -1 $OUT/plain.log)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace*" -delete
echo "[pmc_c5] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- $CMD > $OUT/pmc_fetch.log 2>&1
echo "[pmc_c5] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- $CMD > $OUT/pmc_write.log 2>&1
echo "[pmc_c5] WRITE_SIZE done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -o c -- $CMD > $OUT/pmc_mfma.log 2>&1
echo "[pmc_c5] MFMA counters done"
