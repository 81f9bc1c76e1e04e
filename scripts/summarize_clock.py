"""gpurun_out/<tag>/ (scripts/pmc_clock.sh) -> table of effective clock and MFMA-pipe occupancy per variant.

usage: python scripts/summarize_clock.py <tag>
clock = sum of GRBM_GUI_ACTIVE over the NIF dispatches of the timed step / 8 XCDs / the kernels' HIP-event time printed by
quick_bench.py (MI355X_MICROARCH.md, "DVFS give-back": within 3 % of the in-kernel clock for dispatches >= 10 ms);
MFMA-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
print("%-16s %9s %9s %10s %9s %9s" % ("variant", "NIF TF/s", "nif ms", "clock GHz", "MFMA busy", "Mpath/s"))
for log in sorted(glob.glob(os.path.join(src, "*.log"))):
    name = os.path.basename(log)[:-4]
    line = [l for l in open(log) if l.startswith("variant")]
    if not line:
        continue
    m = re.search(r"Mpaths/s ([\d.]+) NIF TFLOP/s ([\d.]+) trace ms ([\d.]+) nif ms ([\d.]+)", line[-1])
    mpaths, tf, nif_ms = float(m.group(1)), float(m.group(2)), float(m.group(4))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "nif" in row["Kernel_Name"] and (len(sys.argv) < 3 or sys.argv[2] in row["Kernel_Name"]):
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(per)
    half = ids[len(ids) // 2:]          # quick_bench runs a warm-up step and a timed step: the second half is the timed one
    gui = sum(per[i]["GRBM_GUI_ACTIVE"] for i in half)
    busy = sum(per[i]["SQ_VALU_MFMA_BUSY_CYCLES"] for i in half)
    clock = gui / 8.0 / (nif_ms * 1e-3) / 1e9
    print("%-16s %9.1f %9.2f %10.3f %9.3f %9.1f" % (name, tf, nif_ms, clock, busy / (gui / 8.0 * 1024.0), mpaths))
