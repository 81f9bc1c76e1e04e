"""gpurun_out/<tag>/ (scripts/pmc_trace.sh) -> profiles/<tag>_trace_pmc.json: per-path counters of trace_kernel.

usage: python scripts/summarize_trace_pmc.py <tag>
Totals over every trace_kernel dispatch of the run divided by the paths the run traced (trace_bench.py: two steps of
1104 x 1000 x 64 samples).  HBM bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB, separate passes;
gfx950 reports half the bytes of wide coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024 (the kernel's reads are the
4-byte pixel words, its writes 1- and 4-byte scattered stores: neither width is calibrated, so the byte figure is indicative).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATHS = 2 * 64 * 1104 * 1000


def totals(folder, kernel="trace_kernel"):
    out = collections.defaultdict(float)
    n = set()
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"] and "paths" not in row["Kernel_Name"]:
                out[row["Counter_Name"]] += float(row["Counter_Value"])
                n.add(row["Dispatch_Id"])
    return dict(out), len(n)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    depth = int(open(os.path.join(src, "depth")).read()) if os.path.exists(os.path.join(src, "depth")) else 8
    sq, n = totals(os.path.join(src, "sq"))
    fetch, _ = totals(os.path.join(src, "fetch"))
    write, _ = totals(os.path.join(src, "write"))
    doc = {"source": "scripts/pmc_trace.sh %s: rocprofv3 --pmc passes over scripts/trace_bench.py %d (constant sky, 1104x1000, 2 x 64 spp, "
                     "depth %d); totals over %d trace_kernel dispatches / %d paths" % (tag, depth, depth, n, PATHS),
           "max_path_length": depth,
           "paths": PATHS, "dispatches": n, "totals": sq}
    if "SQ_INSTS_VALU" in sq:
        doc["valu_wave_instr_per_path"] = sq["SQ_INSTS_VALU"] / PATHS          # wave-level instructions (64 lanes each)
        doc["valu_lane_instr_per_path_if_full"] = 64.0 * sq["SQ_INSTS_VALU"] / PATHS
    if sq.get("SQ_ACTIVE_INST_VALU") and sq.get("SQ_THREAD_CYCLES_VALU"):
        # SQ_THREAD_CYCLES_VALU counts active lanes per VALU cycle; SQ_ACTIVE_INST_VALU the (quad-)cycles VALU instructions execute
        doc["lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
    if sq.get("GRBM_GUI_ACTIVE") and sq.get("SQ_ACTIVE_INST_VALU"):
        doc["gpu_cycles_total"] = sq["GRBM_GUI_ACTIVE"] / 8.0
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
        doc["valu_busy_fraction"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (sq["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if sq.get("SQ_INSTS_VALU"):
            doc["cycles_per_valu_wave_instr"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
    if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        doc["FETCH_SIZE_KiB"] = fetch["FETCH_SIZE"]
        doc["WRITE_SIZE_KiB"] = write["WRITE_SIZE"]
        doc["hbm_bytes_per_path"] = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0 / PATHS
    json.dump(doc, open(os.path.join(ROOT, "profiles", tag + "_trace_pmc.json"), "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
