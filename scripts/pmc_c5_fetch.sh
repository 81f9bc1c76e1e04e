#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_c5_fetch.sh <tag> [PTMI_GEMM_DIAG value]
# HBM traffic of the C5 layer kernel for one build variant: FETCH_SIZE and WRITE_SIZE in separate counter-only passes over
# scripts/bench_c5.py 8 (a PTMI_GEMM_DIAG value selects a variant of the profiling build; none = the product library).
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then export PTMI_GEMM_DIAG=$2; fi
CMD="python3 $ROOT/scripts/bench_c5.py 8"
$CMD > $OUT/plain.log 2>&1
echo "[pmc_c5_fetch $1] plain: $(tail -1 $OUT/plain.log)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- $CMD > $OUT/pmc_write.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
def med(folder, name):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(out, folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                d[row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ptd::", "")][row["Dispatch_Id"]] += float(row["Counter_Value"])
    res = {}
    for k, v in d.items():
        vals = sorted(v.values()); big = [x for x in vals if x > 0.5 * vals[-1]] or vals
        res[k] = big[len(big) // 2]
    return res
f, w = med("pmc_fetch", "FETCH_SIZE"), med("pmc_write", "WRITE_SIZE")
for k in sorted(f):
    if k.startswith("nifg16_layer"):
        print("[pmc_c5_fetch] %-32s FETCH x2 %.1f MB  WRITE %.1f MB  total %.1f MB" % (k, 2 * f[k] * 1024 / 1e6, w.get(k, 0) * 1024 / 1e6, (2 * f[k] + w.get(k, 0)) * 1024 / 1e6))
PY
