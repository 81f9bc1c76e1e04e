"""gpurun_out/<tag>/ (scripts/pmc_c5.sh) -> profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json for config C5.

usage: python scripts/summarize_c5.py <tag>
HBM bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KiB, separate passes; on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads (the layer kernel's loads are 16 B per lane), so read bytes = 2 x FETCH_SIZE x 1024.
Per kernel: median over the full-size dispatches, beside the ALGORITHMIC bytes of a full chunk (4096 tiles of 32 samples,
hidden 1024: activations in = out = 4096 x 64 KiB = 256 MiB per hidden layer, + 2 MiB of weights)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(folder):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Kernel_Name"]][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: list(d.values()) for c, d in v.items()} for k, v in out.items()}


def median_of_full(vals):
    vals = sorted(vals)
    big = [v for v in vals if v > 0.5 * vals[-1]] or vals
    return big[len(big) // 2]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    avg_ns = {}
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
        for row in csv.DictReader(open(stats[0])):
            avg_ns[row["Name"]] = float(row["AverageNs"])
    fetch, write, mfma = (counters(os.path.join(src, d)) for d in ("pmc_fetch", "pmc_write", "pmc_mfma"))
    MiB = 1 << 20
    algorithmic = {"nifg16_layer_kernel<0": ("hidden layer of a full chunk: 256 MiB activations in + 256 MiB out + 2 MiB weights", 514 * MiB),
                   "nifg16_layer_kernel<1": ("last hidden layer of a full chunk, head fused: 256 MiB activations in + 2 MiB weights + 16 MiB of head partial sums out", 274 * MiB),
                   "nifg16_finish_kernel": ("16 MiB of head partial sums + 1 MiB of queue coordinates in + 1.5 MiB of results out", int(18.5 * MiB)),
                   "nifg16_encode_kernel": ("1 MiB of queue coordinates in + 16 MiB of feature pieces out", 17 * MiB),
                   "nifg_layer_kernel": ("(round-2 kernel) hidden layer of a full chunk: 256 MiB activations in + 256 MiB out + 2 MiB weights", 514 * MiB),
                   "nifg_head_kernel": ("(round-2 kernel) 256 MiB activations in + 1.5 MiB of results", int(257.5 * MiB)),
                   "nifg_encode_kernel": ("(round-2 kernel) 1 MiB of queue coordinates in + 12 MiB of feature pieces out", 13 * MiB)}
    doc = {"source": "scripts/pmc_c5.sh %s over scripts/bench_c5.py 8 (1104x1000, NIF 8x1024, 8 spp); median over the full-size "
                     "dispatches of each kernel; layer 0 (48 -> 1024) and the hidden layers share nifg16_layer_kernel<0, 0>: the median "
                     "is a hidden layer; nifg16_layer_kernel<1, 0> is the last hidden layer with the head fused" % tag,
           "units": "FETCH_SIZE/WRITE_SIZE in KiB; gfx950 correction: read bytes = 2 x FETCH_SIZE x 1024",
           "plain_run": open(os.path.join(src, "plain.log")).read().strip().splitlines()[-1]}
    for kern in sorted(set(fetch) | set(write) | set(mfma)):
        short = kern.split("(")[0].replace("void ", "").replace("ptd::", "")
        e = {}
        if kern in fetch and "FETCH_SIZE" in fetch[kern]:
            e["FETCH_SIZE_KiB"] = median_of_full(fetch[kern]["FETCH_SIZE"])
        if kern in write and "WRITE_SIZE" in write[kern]:
            e["WRITE_SIZE_KiB"] = median_of_full(write[kern]["WRITE_SIZE"])
        if "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
            e["hbm_bytes_per_launch_corrected"] = int(2 * e["FETCH_SIZE_KiB"] * 1024 + e["WRITE_SIZE_KiB"] * 1024)
        for key, (what, nbytes) in algorithmic.items():
            if short.startswith(key) and "hbm_bytes_per_launch_corrected" in e:
                e["algorithmic_bytes_per_launch"] = nbytes
                e["algorithmic_what"] = what
                e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch_corrected"] / nbytes
        for name, ns in avg_ns.items():
            if name.split("(")[0].replace("void ", "").replace("ptd::", "") == short:
                e["avg_launch_us_kernel_trace"] = ns / 1e3     # average over ALL dispatches, short ones included
                if "hbm_bytes_per_launch_corrected" in e and short.startswith(("nifg_head", "nifg16_finish")):
                    e["hbm_TBps"] = e["hbm_bytes_per_launch_corrected"] / (ns * 1e-9) / 1e12
        if kern in mfma and "SQ_VALU_MFMA_BUSY_CYCLES" in mfma[kern]:
            m = {c: median_of_full(v) for c, v in mfma[kern].items()}
            e.update(m)
            if m.get("GRBM_GUI_ACTIVE"):
                e["mfma_pipe_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
                e["gpu_cycles_per_launch"] = m["GRBM_GUI_ACTIVE"] / 8
        if e:
            doc[short] = e
    json.dump(doc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
    print(json.dumps(doc, indent=1)[:6000])


if __name__ == "__main__":
    main()
