#!/usr/bin/env python3
"""The BAL 9-parameter path (ba_solve_bal, csrc/ba_bal.hpp) at BASELINE config 5's size: the BAL-like chain of
synthetic.make_bal_like (1723 cameras / 156 502 points / ~679 k observations) written as a BAL problem (one focal length,
k1 = k2 = 0 at the start, all three adjusted per camera), next to the same data through the tuned 6-parameter path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend                              # noqa: E402
from bundle_adjustment_amd.bal import from_pinhole                         # noqa: E402
from bundle_adjustment_amd.synthetic import make_bal_like                  # noqa: E402

small = len(sys.argv) > 1 and sys.argv[1] == "small"
K4 = np.array([900.0, 900.0, 640.0, 360.0])
pin = make_bal_like(200, 18000, 78000, seed=0, K4=K4) if small else make_bal_like(seed=0, K4=K4)
bal = from_pinhole(pin)
kw = dict(loss="huber", max_iters=50, ftol=1e-5, xtol=1e-5, gtol=1e-8, pcg_tol=0.1, pcg_max_iters=200)
with hip_backend.Solver(0) as s:
    for pre in ("jacobi", "schur_jacobi"):
        for rep in range(2):
            t = time.perf_counter()
            out, cams, pts = s.solve_bal(bal, fixed_cam=0, preconditioner=pre, **kw)
            dt = time.perf_counter() - t
        print(f"BAL 9-parameter path ({pre}): {bal.n_cams} cams / {bal.n_pts} pts / {bal.n_obs} obs: {out['iterations']} LM iterations, "
              f"{out['pcg_iterations']} PCG iterations, RMSE {np.sqrt(out['initial_sse'] / bal.n_obs):.3f} -> "
              f"{np.sqrt(out['final_sse'] / bal.n_obs):.4f} px, {out['status_name']}, solve {out['seconds_total'] * 1e3:.1f} ms "
              f"({out['iterations'] / out['seconds_total']:.0f} LM it/s; {dt * 1e3:.1f} ms with upload)")
        print(f"   per PCG iteration {out['seconds_pcg'] / max(out['pcg_iterations'], 1) * 1e6:.0f} us; f moved by up to "
              f"{np.abs(cams[:, 6] / bal.cams[:, 6] - 1).max() * 100:.3f} %, |k1| up to {np.abs(cams[:, 7]).max():.2e}")
    for pre in ("jacobi", "schur_jacobi"):
        s.set_problem(pin)
        for rep in range(2):
            s.set_params(pin.cams, pin.pts)
            o6 = s.solve(preconditioner=pre, **kw)
        print(f"6-parameter path ({pre}, same tolerances): {o6['iterations']} LM iterations, {o6['pcg_iterations']} PCG iterations, "
              f"RMSE -> {np.sqrt(o6['final_sse'] / pin.n_obs):.4f} px, solve {o6['seconds_total'] * 1e3:.1f} ms; per PCG iteration "
              f"{o6['seconds_pcg'] / max(o6['pcg_iterations'], 1) * 1e6:.0f} us")
