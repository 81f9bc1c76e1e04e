"""Turn gpurun_out/<tag>/ (scripts/profile_round.sh) into the summaries committed under profiles/.

usage: python scripts/summarize_profile.py <tag> [destination = profiles/]
Writes <destination>/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc.json.  scripts/profile_round.sh runs it ON THE GPU BOX
with gpurun_out/<tag>/summary as destination and then deletes the raw counter CSVs (tens of MB per pass; gpurun copies back
at most 64 MiB): commit the summary files under profiles/.  HBM bytes follow MI355X_MICROARCH.md's
HBM/rocprofv3 section: FETCH_SIZE and WRITE_SIZE are KiB, collected in separate passes; on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(folder):
    """{kernel: {counter: [per-dispatch sums]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Kernel_Name"]][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: list(d.values()) for c, d in v.items()} for k, v in out.items()}


def median_of_full(vals):
    vals = sorted(vals)
    big = [v for v in vals if v > 0.5 * vals[-1]] or vals     # full-size launches only (the last batch of a step is short)
    return big[len(big) // 2]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    pmc = {"source": "scripts/profile_round.sh %s: separate rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0`; "
                     "median over the full-size launches of each kernel" % tag,
           "units": "FETCH_SIZE/WRITE_SIZE are KiB; gfx950 correction: read bytes = 2 x FETCH_SIZE x 1024 "
                    "(MI355X_MICROARCH.md, HBM / rocprofv3)"}
    fetch, write, mfma = (counters(os.path.join(src, d)) for d in ("pmc_fetch", "pmc_write", "pmc_mfma"))
    for kern in sorted(set(fetch) | set(write) | set(mfma)):
        short = kern.split("(")[0].replace("void ", "").replace("ptd::", "")
        e = {}
        if kern in fetch and "FETCH_SIZE" in fetch[kern]:
            e["FETCH_SIZE_KiB"] = median_of_full(fetch[kern]["FETCH_SIZE"])
        if kern in write and "WRITE_SIZE" in write[kern]:
            e["WRITE_SIZE_KiB"] = median_of_full(write[kern]["WRITE_SIZE"])
        if "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
            e["hbm_bytes_per_launch_corrected"] = int(2 * e["FETCH_SIZE_KiB"] * 1024 + e["WRITE_SIZE_KiB"] * 1024)
        if kern in mfma and "SQ_VALU_MFMA_BUSY_CYCLES" in mfma[kern]:
            m = {c: median_of_full(v) for c, v in mfma[kern].items()}
            e.update(m)
            if m.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs on the chip
                e["mfma_pipe_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if e:
            pmc[short] = e
    json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
    json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in pmc.items() if isinstance(v, dict)}, indent=1)[:3000])


if __name__ == "__main__":
    main()
