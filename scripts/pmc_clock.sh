#!/bin/bash
# usage (GPU box): scripts/pmc_clock.sh <tag> <name>=<ENV:VAL,...> ...
# For each variant of the profiling build one rocprofv3 --pmc pass (counters only) over scripts/quick_bench.py 120:
# GRBM_GUI_ACTIVE (sum over 8 XCDs) and the MFMA-pipe counters per NIF dispatch, and the NIF kernels' own HIP-event
# time printed by the script.  scripts/summarize_clock.py <tag> divides them: effective clock, MFMA-pipe busy fraction.
set -e
ROOT=$GRAFT_REPO_ROOT; TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export QB_DIAG=1   # quick_bench.py loads libptmi_diag.so
for v in "$@"; do
  name=${v%%=*}; envs=${v#*=}
  unset PTMI_NIF_DIAG PTMI_NIF_VARIANT PTMI_SERIAL PTMI_NIF_KERNEL PTMI_GEMM_DIAG
  IFS=',' read -ra kvs <<< "$envs"
  for kv in "${kvs[@]}"; do [ -n "$kv" ] && export "${kv%%:*}=${kv#*:}"; done
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/$name -o c -- python3 $ROOT/scripts/quick_bench.py ${SPP:-120} > $OUT/$name.log 2>&1
  echo "[pmc_clock] $name done: $(grep variant $OUT/$name.log | tail -1)"
done
