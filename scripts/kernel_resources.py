#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of the product library (hipcc -Rpass-analysis=kernel-resource-usage).

usage: python scripts/kernel_resources.py [filter-substring] [-DPTMI_DIAG_BUILD ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("-")]
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-o", "/tmp/_ptmi_res.so",
           os.path.join(ROOT, "ipu_path_trace_amd", "csrc", "ptmi.hip"), "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": name}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[bytes/block\])?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for r in rows:
        if flt and not any(f in r["name"] for f in flt):
            continue
        print("%-90s VGPR %3d AGPR %3d SGPR %3d spillV %d spillS %d LDS %6d occ %s" % (
            r["name"][:90], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs", -1), r.get("VGPRs Spill", -1),
            r.get("SGPRs Spill", -1), r.get("LDS Size [bytes/block]", -1), r.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
