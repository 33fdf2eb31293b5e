#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes, as
MI355X_MICROARCH.md prescribes) into profiles/traffic.json + a per-kernel CSV.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts the
64-byte half of every 128-byte request, so it is doubled (guide, section HBM); launches that
exit early (converged PCG) are excluded by taking the median over launches above 20 % of the max.
"""
import collections
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the PCG instantiations: k_pt_schur[_both]<CM, ROBUST, MODE 0, ...>, k_cam_schur<CM, ROBUST, PCG true, ...>
import re
SLOT = [(re.compile(r"k_pt_schur(_both)?<ba::\w+, (true|false), 0,"), "schur_pt"),
        (re.compile(r"k_cam_schur<ba::\w+, (true|false), true,"), "schur_cam"),
        (re.compile(r"k_pt_linearize(_both)?<"), "linearize_pt"), (re.compile(r"k_camrow_linearize<"), "linearize_cam"),
        (re.compile(r"k_camrow_schur_diag<"), "precond"), (re.compile(r"k_pt_schur(_both)?<ba::\w+, (true|false), 1,"), "backsub_pt")]


def agg(path, cname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def med(vals):
    vals = [v for v in vals if v > 0.2 * max(vals)] if vals and max(vals) > 0 else [0.0]
    return statistics.median(vals)


def main():
    f, w, tag = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    rows, traffic, launches = [], {}, {}
    for k in sorted(f):
        fk, wk = med(f[k]), med(w.get(k, [0.0]))
        hbm = (2 * fk + wk) * 1024
        rows.append((k, len(f[k]), fk, wk, hbm))
        for pat, slot in SLOT:
            if pat.search(k) and len(f[k]) > launches.get(slot, 0):        # several instantiations of a slot: the one the loop launches most
                traffic[slot], launches[slot] = round(hbm), len(f[k])
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_per_kernel.csv"), "w") as out:
        out.write("kernel,launches,FETCH_SIZE_KB_median,WRITE_SIZE_KB_median,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
        for k, n, fk, wk, hbm in rows:
            out.write(f"\"{k}\",{n},{fk:.1f},{wk:.1f},{hbm:.0f}\n")
    traffic["_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command, profile tag {tag} (tools/profile_round.sh)"
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(traffic)


if __name__ == "__main__":
    main()
