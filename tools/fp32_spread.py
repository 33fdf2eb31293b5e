import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend
from tests.test_bal import _synthetic_bal
p = _synthetic_bal(40, 2000, 5, seed=12)
kw = dict(fixed_cam=0, loss="huber", max_iters=60, ftol=1e-12, xtol=1e-12, gtol=0.0, pcg_tol=1e-2, pcg_max_iters=400)
runs = {}
with hip_backend.Solver(0) as s:
    runs["f64"] = s.solve_bal(p, **kw)
    runs["f32"] = s.solve_bal(p, jacobian_precision=1, **kw)
    runs["f64 lam0 1e-3"] = s.solve_bal(p, initial_lambda=1e-3, **kw)
    runs["f64 pcg 1e-3"] = s.solve_bal(p, **dict(kw, pcg_tol=1e-3))
    runs["f64 lag0 model0"] = s.solve_bal(p, precond_lag=0, pcg_model_tol=0.0, **kw)
a, ca, pa = runs["f64"]
for k, (b, cb, pb) in runs.items():
    sc = np.linalg.norm(ca[1:, 3:6]) / np.linalg.norm(cb[1:, 3:6])
    print(f"{k:18s} its {b['iterations']:3d} {b['status_name']:9s} cost rel {abs(a['final_cost'] - b['final_cost']) / a['final_cost']:.2e} rot {np.abs(ca[:, :3] - cb[:, :3]).max():.2e} "
          f"intr rel {np.abs(ca[:, 6:] / cb[:, 6:] - 1).max():.2e} scale-1 {abs(sc - 1):.2e} pts {np.abs(sc * pb - pa).max() / np.abs(pa).max():.2e}")
