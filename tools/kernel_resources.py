#!/usr/bin/env python3
"""VGPR / SGPR / scratch / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "bundle_adjustment_amd", "csrc", "ba_hip.hip")
flt = sys.argv[1:] or [""]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
                    "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_res.so", src, "-ldl", "-lrt"],
                   capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
KEYS = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("TotalSGPRs", "SGPR"), (r"ScratchSize \[bytes/lane\]", "scratch"),
        (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "LDS")]
for b in blocks:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ba::", "")
    if not any(f in name for f in flt):
        continue
    vals = []
    for k, lab in KEYS:
        m = re.search(k + r": (\d+)", b)
        vals.append(f"{lab} {m.group(1) if m else '?'}")
    print(f"{name[:70]:70s} " + "  ".join(vals))
