import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bundle_adjustment_amd import hip_backend
from oracle import ba_oracle as o
from tests.test_bal import _synthetic_bal
p, loss, fixed = _synthetic_bal(20, 600, 5, seed=8), "huber", 0
iters = 8
for lag in (0, 3):
    ref = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, None, fixed_cam=fixed, loss=loss, max_iters=iters, ftol=0.0, xtol=0.0,
                     gtol=0.0, pcg_tol=1e-2, pcg_max_iters=300, precond="schur_jacobi", model="bal", precond_lag=lag)
    with hip_backend.Solver(0) as s:
        out, cams, pts = s.solve_bal(p, fixed_cam=fixed, loss=loss, max_iters=iters, ftol=0.0, xtol=0.0, gtol=0.0, pcg_tol=1e-2,
                                     pcg_max_iters=300, pcg_min_iters=0, preconditioner="schur_jacobi", precond_lag=lag)
        tr = s.trace()
        print("lag", lag, s.stats())
    for t, h in zip(tr, ref["history"]):
        print("  it %2d pcg %3d/%3d acc %d cost_trial %.15e / %.15e lam %.3e/%.3e" % (t["iteration"], t["pcg_iterations"], h["pcg"], t["accepted"], t["cost_trial"], h["cost_new"], t["damping"], h["lam"]))
