#!/bin/bash
# usage (GPU box, via gpurun): scripts/nb2_bound.sh <tag>
# What could an NB = 2 version of the headline kernel buy (two 32-sample tiles per wave sharing every weight fragment: half
# the LDS -> register bytes per MFMA)?  Three measurements, all on the 6x320 network inside the C2 step on one device:
#  1. same-process A/B (scripts/ab_nif.py): the product kernel; the SAME kernel issuing half / none of its LDS fragment reads
#     (timing only: an upper bound, nothing else changes -- a real NB = 2 kernel also pays for moving activations through
#     AGPRs and loses its partner wave); the v2 ring kernel at 8 waves x 32 samples and at 4 waves x 64 samples (a real NB = 2).
#  2. the in-kernel clock of the first three (scripts/clock_nif.py).
#  3. LDS counters of the product kernel and of the half-reads variant (counter-only rocprofv3 passes).
set -e
ROOT=$GRAFT_REPO_ROOT; TAG=$1; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/ab_nif.py 4 150 product= halfreads=PTMI_NIF_DIAG:64 noreads=PTMI_NIF_DIAG:2 v2_8x32=PTMI_NIF_VARIANT:3 v2_4x64_nb2=PTMI_NIF_VARIANT:2 > $OUT/ab.txt 2>&1
echo "[nb2_bound] A/B done"; cat $OUT/ab.txt
python3 $ROOT/scripts/clock_nif.py 3 120 32 96 34 > $OUT/clock.txt 2>&1
echo "[nb2_bound] clocks done"; cat $OUT/clock.txt
export QB_DIAG=1
for v in product:0 halfreads:64; do
  name=${v%%:*}; d=${v#*:}
  if [ "$d" != "0" ]; then export PTMI_NIF_DIAG=$d; else unset PTMI_NIF_DIAG; fi
  i=0
  for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${name}_$i -o c -- python3 $ROOT/scripts/quick_bench.py 120 > $OUT/pmc_${name}_$i.log 2>&1 || echo "[nb2_bound] pass $name/$i failed (a counter of this set may not exist on gfx950)"
    echo "[nb2_bound] pmc $name pass $i done"
  done
done
unset PTMI_NIF_DIAG
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for name in ("product", "halfreads"):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(os.path.join(out, "pmc_%s_*" % name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "nif_kernel_v3" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print("[nb2_bound] %s: NIF dispatches %s" % (name, dict(n)))
    for k in sorted(tot):
        print("[nb2_bound]   %-26s %18.0f  per dispatch %.4g" % (k, tot[k], tot[k] / max(n[k], 1)))
    if tot.get("SQ_BUSY_CYCLES") and tot.get("SQ_ACTIVE_INST_LDS"):
        # SQ_ACTIVE_INST_LDS counts (per SIMD... summed over the chip) cycles in which an LDS instruction is in flight
        print("[nb2_bound]   LDS-instruction-active cycles / SQ busy cycles = %.3f" % (tot["SQ_ACTIVE_INST_LDS"] / tot["SQ_BUSY_CYCLES"]))
    if tot.get("SQ_LDS_IDX_ACTIVE") and tot.get("SQ_LDS_BANK_CONFLICT") is not None:
        print("[nb2_bound]   LDS bank-conflict cycles / LDS active cycles = %.4f" % (tot["SQ_LDS_BANK_CONFLICT"] / max(tot["SQ_LDS_IDX_ACTIVE"], 1)))
PY
