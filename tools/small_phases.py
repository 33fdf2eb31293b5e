#!/usr/bin/env python3
"""Where the single-launch window solver (csrc/ba_small.hpp) spends one LM iteration.

    BA_SMALL_STAMPS=1 python tools/small_phases.py [n_cams n_pts obs_per_pt]

With BA_SMALL_STAMPS set, thread 0 of k_small_lm records the 100 MHz clock at the phase boundaries of LM iteration 2
(device memory; copied back after the solve) and the library prints the phase times on stderr."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend                    # noqa: E402
from bundle_adjustment_amd.synthetic import make_problem         # noqa: E402

n_cams, n_pts, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 500, 4)
p = make_problem(n_cams, n_pts, k, seed=0)
with hip_backend.Solver(0) as s:
    for i in range(3):
        s.set_problem(p)
        out = s.solve(loss="huber")
print(f"{n_cams} cams / {n_pts} pts / {p.n_obs} obs: {out['iterations']} LM iterations, final cost {out['final_cost']:.6f}, "
      f"solve {out['seconds_total'] * 1e3:.3f} ms")
