#!/usr/bin/env python3
"""Where the GPU waits between kernels: from a rocprofv3 --kernel-trace CSV, the idle time between consecutive kernels of
the solver's stream, aggregated by (kernel that ended -> kernel that started).  python tools/gap_summary.py <run_kernel_trace.csv> [min_us]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ba::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:44]))
rows.sort()
gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
busy = 0.0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) * 1e-3
    if g > 200.0:                      # between solves (host work outside the loop)
        continue
    k = (n0, n1)
    gaps[k][0] += 1
    gaps[k][1] += max(g, 0.0)
    gaps[k][2] = max(gaps[k][2], g)
tot = sum(v[1] for v in gaps.values())
print(f"{len(rows)} kernels; idle between consecutive kernels (gaps <= 200 us): {tot:.0f} us in all")
for (a, b), (n, t, mx) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"  {a:44s} -> {b:44s} x{n:5d}  mean {t / n:6.2f} us  max {mx:7.2f}  total {t:8.0f} us")
