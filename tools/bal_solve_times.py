#!/usr/bin/env python3
"""BASELINE config 5 as stated, time to solution: the BAL 9-parameter camera (ba_solve_bal; every kernel of the LM / Schur /
PCG loop instantiated for ba_models.hpp's BalCam) on synthetic.make_bal_problem (1723 cameras / 156 502 points / ~662 k
observations, distinct f / k1 / k2 per camera), fp64 and in the config's precision mode (fp32 Jacobian blocks in the PCG
passes), next to the same chain through the reference's pinhole model (make_bal_like).  python tools/bal_solve_times.py [small]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend                              # noqa: E402
from bundle_adjustment_amd.synthetic import make_bal_like, make_bal_problem   # noqa: E402

small = len(sys.argv) > 1 and sys.argv[1] == "small"
sizes = (200, 18000, 78000) if small else (1723, 156502, 678718)
bal = make_bal_problem(*sizes, seed=0)
pin = make_bal_like(*sizes, seed=0)
kw = dict(loss="huber", max_iters=50, ftol=1e-5, xtol=1e-5, gtol=1e-8, pcg_tol=0.1, pcg_max_iters=200)
with hip_backend.Solver(0) as s:
    intr0 = s.set_problem_bal(bal, fixed_cam=0)
    for pre, jp in (("jacobi", 0), ("schur_jacobi", 0), ("schur_jacobi", 1)):
        for rep in range(2):
            s.set_params(bal.cams[:, :6], bal.pts)
            intr = intr0.copy()
            t = time.perf_counter()
            out = s.solve_bal_resident(intr, preconditioner=pre, jacobian_precision=jp, **kw)
            dt = time.perf_counter() - t
        tr = s.trace()
        print(f"BAL camera ({pre}, {'fp32 Jacobian blocks in the PCG passes' if jp else 'fp64'}): {bal.n_cams} cams / {bal.n_pts} pts / {bal.n_obs} obs: "
              f"{out['iterations']} LM iterations, {out['pcg_iterations']} PCG iterations, RMSE {np.sqrt(out['initial_sse'] / bal.n_obs):.3f} -> "
              f"{np.sqrt(out['final_sse'] / bal.n_obs):.4f} px, {out['status_name']}, solve {out['seconds_total'] * 1e3:.1f} ms "
              f"({out['iterations'] / out['seconds_total']:.0f} LM it/s)")
        print(f"   per PCG iteration {out['seconds_pcg'] / max(out['pcg_iterations'], 1) * 1e6:.1f} us; PCG per LM {[t_['pcg_iterations'] for t_ in tr]}; "
              f"f moved by up to {np.abs(intr[:, 0] / intr0[:, 0] - 1).max() * 100:.3f} %, |k2| up to {np.abs(intr[:, 2]).max():.2e}")
    for pre in ("jacobi", "schur_jacobi"):
        s.set_problem(pin)
        for rep in range(2):
            s.set_params(pin.cams, pin.pts)
            o6 = s.solve(preconditioner=pre, **kw)
        print(f"pinhole model ({pre}, same tolerances): {o6['iterations']} LM iterations, {o6['pcg_iterations']} PCG iterations, "
              f"RMSE -> {np.sqrt(o6['final_sse'] / pin.n_obs):.4f} px, solve {o6['seconds_total'] * 1e3:.1f} ms; per PCG iteration "
              f"{o6['seconds_pcg'] / max(o6['pcg_iterations'], 1) * 1e6:.1f} us")
