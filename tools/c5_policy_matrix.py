#!/usr/bin/env python3
"""BASELINE config 5 on the BAL camera (1723 cameras / 156 502 points / ~662 k observations): what the two round-4 policies
for ill-conditioned inner solves buy -- the cap-aware damping floor (an inner solve that runs into pcg_max_iters keeps the
damping from falling further; BA_NO_CAP_FLOOR=1 switches it off) and the PCG model test (ba_options.pcg_model_tol; automatic
on band-structured problems at loose outer tolerances) -- at the reference's tolerances (src/bundle_adjuster.py:170-174:
ftol = xtol = 1e-5) and at a tight one (ftol = 1e-7).  One subprocess per setting (the floor switch is read per solve).
python tools/c5_policy_matrix.py"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.synthetic import make_bal_problem
    bal = make_bal_problem(1723, 156502, 678718, seed=0)
    mt = float(sys.argv[1])
    for label, kw in (("reference tolerances (ftol 1e-5, cap 200)", dict(loss="huber", max_iters=50, ftol=1e-5, xtol=1e-5, gtol=1e-8, pcg_tol=0.1, pcg_max_iters=200)),
                      ("tight (ftol 1e-7, cap 300, 60 iterations)", dict(loss="huber", max_iters=60, ftol=1e-7, xtol=1e-10, gtol=1e-10, pcg_tol=0.1, pcg_max_iters=300))):
        with hip_backend.Solver(0) as s:
            intr0 = s.set_problem_bal(bal, fixed_cam=0)
            for rep in range(2):
                s.set_params(bal.cams[:, :6], bal.pts)
                intr = intr0.copy()
                out = s.solve_bal_resident(intr, pcg_model_tol=mt, **kw)
            tr = s.trace()
        print(f"floor {'off' if os.environ.get('BA_NO_CAP_FLOOR') else 'on '}  model test {mt:3.1f}  [{label}]: {out['iterations']:2d} LM, {out['pcg_iterations']:5d} PCG, "
              f"RMSE {np.sqrt(out['final_sse'] / bal.n_obs):.6f} px, {out['status_name']:9s} {out['seconds_total'] * 1e3:6.1f} ms   PCG per LM {[t['pcg_iterations'] for t in tr]}", flush=True)
else:
    for floor in ("off", "on"):
        for mt in ("0", "0.5"):
            env = dict(os.environ)
            if floor == "off":
                env["BA_NO_CAP_FLOOR"] = "1"
            subprocess.run([sys.executable, os.path.abspath(__file__), mt], env=env)
