#!/bin/bash
# usage: scripts/pmc_stalls.sh <tag>   (GPU box): stall-attribution counters for the C5 layer kernel and the C2 NIF kernel,
# three --pmc passes each (counters only).
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
B="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
C="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_SALU"
i=0
for set in "$A" "$B" "$C"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/c5_$i -o c -- python3 $ROOT/scripts/bench_c5.py 4 > $OUT/c5_$i.log 2>&1
  rocprofv3 --pmc $set --output-format csv -d $OUT/c2_$i -o c -- python3 $ROOT/scripts/quick_bench.py 32 > $OUT/c2_$i.log 2>&1
  echo "[pmc_stalls] pass $i done"
done
