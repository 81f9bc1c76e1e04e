"""gpurun_out/<tag>/ (scripts/pmc_stalls.sh) -> profiles/<tag>_pmc.json: per-wave-cycle stall split of the C5 layer kernel and
the C2 fused NIF kernel.  usage: python scripts/summarize_stalls.py <tag>
Counters are summed over the full-size dispatches of each kernel and divided by SQ_WAVE_CYCLES (cycles a wave is resident)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(folder):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ptd::", "")
            per[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    out = {}
    for k, cs in per.items():
        out[k] = {}
        for c, d in cs.items():
            vals = sorted(d.values())
            big = [v for v in vals if v > 0.5 * vals[-1]] or vals
            out[k][c] = big[len(big) // 2]
    return out


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    doc = {"source": "scripts/pmc_stalls.sh %s: three counter-only rocprofv3 passes each over scripts/bench_c5.py 4 (C5, NIF 8x1024) and "
                     "scripts/quick_bench.py 32 (C2, NIF 6x320); median over the full-size dispatches of a kernel" % tag}
    for cfg, want in (("c5", ("nifg16_layer_kernel<0", "nifg16_layer_kernel<1")), ("c2", ("nif_kernel_v3",))):
        merged = collections.defaultdict(dict)
        for i in (1, 2, 3):
            for k, cs in collect(os.path.join(src, "%s_%d" % (cfg, i))).items():
                merged[k].update(cs)
        for k, cs in merged.items():
            if not k.startswith(want):
                continue
            e = dict(cs)
            wc = cs.get("SQ_WAVE_CYCLES")
            if wc:
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                          "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_SALU"):
                    if c in cs:
                        e[c + "_per_wave_cycle"] = cs[c] / wc
            doc[k] = e
    json.dump(doc, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
    for k, e in doc.items():
        if isinstance(e, dict):
            print(k, {c: round(v, 4) for c, v in e.items() if c.endswith("per_wave_cycle")})


if __name__ == "__main__":
    main()
