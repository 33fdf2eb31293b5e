"""Time to solution of the BASELINE configs (not the bench's fixed-iteration rate): one ba_solve with the reference's
own stopping tolerances (loss='huber', xtol = ftol = 1e-5, at most 50 evaluations, src/bundle_adjuster.py:170-174) and
one driven to convergence; LM / PCG iteration counts, wall time of the solve loop, final RMSE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import hip_backend as hb
from bundle_adjustment_amd.synthetic import make_bal_like, make_config

for cfg in (sys.argv[1:] or ["C2", "C3", "C5"]):
    p = make_bal_like(seed=0) if cfg == "C5" else make_config(cfg, seed=0)
    with hb.Solver(0) as s:
        for label, kw in (("reference tolerances", dict()),
                          ("to convergence", dict(max_iters=100, ftol=1e-12, xtol=1e-12, gtol=0.0, pcg_tol=1e-2, pcg_max_iters=1000))):
            for pc in (("schur_jacobi", "two_level") if cfg == "C5" else ("schur_jacobi",)):
                s.set_problem(p)
                s.solve(preconditioner=pc, **kw)               # warm (first launches of each kernel)
                s.set_problem(p)
                out = s.solve(preconditioner=pc, **kw)
                tr = s.trace()
                print(f"{cfg} {p.n_cams}/{p.n_pts}/{p.n_obs} {label:22s} {pc:13s}: {out['iterations']:3d} LM it ({out['accepted']} accepted), "
                      f"{out['pcg_iterations']:5d} PCG it, {1e3 * out['seconds_total']:8.2f} ms, status {out['status_name']}, "
                      f"RMSE {np.sqrt(out['initial_sse'] / p.n_obs):.3f} -> {np.sqrt(out['final_sse'] / p.n_obs):.6f} px, "
                      f"PCG per LM {[t['pcg_iterations'] for t in tr][:12]}", flush=True)
