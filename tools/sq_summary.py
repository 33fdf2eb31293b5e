#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc SQ counter passes (working launches only: >= half the kernel's largest value).
usage: python tools/sq_summary.py <pass1_counter_collection.csv> [<pass2_counter_collection.csv> ...]"""
import collections
import csv
import re
import sys


def table(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ba::", "").replace("ba::", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


print("rocprofv3 --pmc <8 SQ counters> --kernel-trace (one pass per block below), python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 3")
print("C3 (1000 cams / 100k pts / 1M obs); mean over working launches (>= half the kernel's largest value); "
      "SQ_*CYCLES / WAIT / ACTIVE count quad-cycles summed over all waves\n")
for path in sys.argv[1:]:
    d = table(path)
    names = sorted({c for k in d for c in d[k]})
    print(f"{'kernel':58s} " + " ".join(f"{n:>20s}" for n in names) + "  launches")
    for k in sorted(d):
        if not k.startswith("k_"):
            continue
        row = []
        n_l = 0
        for n in names:
            v = d[k].get(n, [0.0])
            mx = max(v) if v else 0.0
            w = [x for x in v if x >= 0.5 * mx] if mx > 0 else v
            row.append(sum(w) / max(len(w), 1))
            n_l = max(n_l, len(v))
        print(f"{k[:58]:58s} " + " ".join(f"{x:20.4g}" for x in row) + f"  {n_l}")
    print()
