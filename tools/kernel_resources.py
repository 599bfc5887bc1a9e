#!/usr/bin/env python3
"""profiles/<tag>_kernel_resources.txt: the compiler's own resource table (-Rpass-analysis=kernel-resource-usage) of every kernel
of pocketsphinx_amd/csrc -- registers, spills, scratch, occupancy, static LDS -- since rocprofv3's VGPR_Count / LDS_Block_Size
columns report allocation granules and static LDS only.   usage: tools/kernel_resources.py <tag>"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pocketsphinx_amd.capi import FILE_FLAGS  # noqa: E402  (the product build's per-source flags: the tree search is built -Os)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-w", "-I" + os.path.join(ROOT, "include"),
         "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "-o", "/dev/null"]
KEYS = [("VGPRs", r"\s+VGPRs: (\d+)"), ("AGPRs", r"AGPRs: (\d+)"), ("SGPRs", r"TotalSGPRs: (\d+)"), ("spillV", r"VGPRs Spill: (\d+)"),
        ("spillS", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
        ("lds", r"LDS Size \[bytes/block\]: (\d+)")]


def main():
    tag = sys.argv[1]
    lines = ["# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage (+ the product build's per-source "
             "flags: %s), every kernel of pocketsphinx_amd/csrc (tools/kernel_resources.py %s)" % (FILE_FLAGS, tag),
             "# (static LDS only: fwdtree_kernel's LDS layout adds 60.8 KB of dynamic LDS reading rows / 65.4 KB scoring from lists, the slab layouts "
             "18 x work-items x 4 B + 64 + the listed-nodes bitmap; fwdflat_kernel's scoring form 4 x n_sen bytes: score row and listed senones)"]
    for src in sorted(glob.glob(os.path.join(ROOT, "pocketsphinx_amd", "csrc", "*.hip"))):
        p = subprocess.run(["hipcc"] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + [src], capture_output=True, text=True, timeout=1800)
        cur = None
        rows = {}
        for ln in p.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", ln)
            if m:
                cur = rows.setdefault(m.group(1), {})
                continue
            if cur is None:
                continue
            for k, pat in KEYS:
                m = re.search(r"remark:" + (pat if pat.startswith(r"\s") else r"\s+" + pat), ln)
                if m:
                    cur[k] = int(m.group(1))
        for name, v in rows.items():
            lines.append("%-22s VGPRs %3d  AGPRs %d  SGPRs %3d  spilled V %3d S %3d  scratch %4d B/lane  waves/SIMD %d  static LDS %6d B  %s" % (
                os.path.basename(src), v.get("VGPRs", -1), v.get("AGPRs", 0), v.get("SGPRs", -1), v.get("spillV", 0), v.get("spillS", 0),
                v.get("scratch", 0), v.get("occ", -1), v.get("lds", 0), name))
    open(os.path.join(ROOT, "profiles", "%s_kernel_resources.txt" % tag), "w").write("\n".join(lines) + "\n")
    print(len(lines) - 2, "kernels")


if __name__ == "__main__":
    main()
