#!/usr/bin/env python3
"""Single-launch window solver vs the multi-kernel path as the window grows: where should ba_solve stop dispatching
to csrc/ba_small.hpp?  Prints solve() wall time (median of 7) and LM iterations for both on the same problems."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend                    # noqa: E402
from bundle_adjustment_amd.synthetic import make_problem         # noqa: E402

os.environ["BA_SMALL_MAX_OBS"] = "1000000"        # lift the dispatch limit for the measurement
with hip_backend.Solver(0) as s:
    for n_cams, n_pts, k in [(5, 250, 4), (5, 500, 4), (5, 1000, 4), (8, 1000, 5), (8, 2000, 5), (8, 4000, 5), (8, 8000, 5)]:
        p = make_problem(n_cams, n_pts, k, seed=0)
        row = [f"{n_cams} cams {n_pts:5d} pts {p.n_obs:6d} obs:"]
        for small in (0, 1):
            ts = []
            for rep in range(8):
                s.set_problem(p)
                t = time.perf_counter()
                out = s.solve(loss="huber", small_solver=small)
                ts.append(time.perf_counter() - t)
            row.append(f"{'single launch' if small == 0 else 'multi-kernel '} {np.median(ts[1:]) * 1e3:7.3f} ms ({out['iterations']:2d} LM it, cost {out['final_cost']:.4f})")
        print("   ".join(row))
