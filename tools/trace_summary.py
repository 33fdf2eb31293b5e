#!/usr/bin/env python3
"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv compactly: calls, mean / min / max us, share."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ba::", "").replace("ba::", "")
    print(f"{n[:50]:50s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:7.2f} us  min {float(r['MinNs']) / 1e3:6.2f}  "
          f"max {float(r['MaxNs']) / 1e3:6.2f}  total {float(r['TotalDurationNs']) / 1e3:8.1f} us  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
