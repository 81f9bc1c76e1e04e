#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_acc.sh <tag>
# accumulate4_kernel alone (profiling build, every kernel on one stream, constant sky at the C2 image): times from
# scripts/acc_bench.py, HBM bytes per launch from two counter-only passes (FETCH_SIZE, WRITE_SIZE: KiB; gfx950 reads x 2).
set -e
ROOT=$GRAFT_REPO_ROOT; TAG=$1; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/acc_bench.py c2 > $OUT/acc_c2.txt 2>&1; cat $OUT/acc_c2.txt
python3 $ROOT/scripts/acc_bench.py c3 1000 2 > $OUT/acc_c3.txt 2>&1; cat $OUT/acc_c3.txt
export PTMI_SERIAL=1 QB_SERIAL_CONST=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o c -- python3 $ROOT/scripts/trace_bench.py 8 diag > $OUT/pmc_$c.log 2>&1 || echo "[pmc_acc] $c pass failed"
  echo "[pmc_acc] $c done"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "accumulate" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot):
    print("[pmc_acc] %-12s %14.0f KiB over %d launches" % (k, tot[k], n[k]))
if tot.get("FETCH_SIZE") and tot.get("WRITE_SIZE"):
    per = (2 * tot["FETCH_SIZE"] / n["FETCH_SIZE"] + tot["WRITE_SIZE"] / n["WRITE_SIZE"]) * 1024
    print("[pmc_acc] HBM bytes per accumulate launch (2 x FETCH + WRITE): %.1f MB" % (per / 1e6))
PY
