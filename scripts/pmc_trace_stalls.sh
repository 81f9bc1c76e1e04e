#!/bin/bash
# usage: scripts/pmc_trace_stalls.sh <tag> [depth = 8]   (GPU box): where the trace kernel's wave-cycles go -- three counter-only
# passes over scripts/trace_bench.py <depth> (constant sky: the kernel on its own).  The depth is recorded in <out>/depth, as
# scripts/pmc_trace.sh does.
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DEPTH=${2:-8}
echo $DEPTH > $OUT/depth
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU"
C="SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH"
i=0
for set in "$A" "$B" "$C"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/t_$i -o c -- python3 $ROOT/scripts/trace_bench.py $DEPTH > $OUT/t_$i.log 2>&1 || echo "[pmc_trace_stalls] pass $i failed (a counter of this set may not exist)"
  echo "[pmc_trace_stalls] pass $i done"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(os.path.join(out, "t_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"] and "paths" not in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
wc = tot.get("SQ_WAVE_CYCLES", 0)
for k in sorted(tot):
    print("[pmc_trace_stalls] %-24s %16.0f  per wave-cycle %.4f" % (k, tot[k], tot[k] / wc if wc else 0))
PY
