import sys
sys.path.insert(0, '.')
import numpy as np
from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.synthetic import make_bal_like
for n_cams in (400, 1000, 1200, 1723):
    scale = n_cams / 1723.0
    p = make_bal_like(n_cams=n_cams, n_pts=int(156502 * scale), n_obs_target=int(678718 * scale), seed=0)
    with hip_backend.Solver(0) as s:
        res = {}
        for pc in ("schur_jacobi", "two_level"):
            s.set_problem(p)
            out = s.solve(loss="huber", max_iters=8, ftol=0.0, xtol=0.0, gtol=0.0, pcg_tol=0.1, pcg_max_iters=1000, preconditioner=pc)
            res[pc] = ([t["pcg_iterations"] for t in s.trace()], out["final_cost"], out["seconds_total"])
        print(n_cams, "SJ", res["schur_jacobi"][0], "%.6e %.4f" % res["schur_jacobi"][1:], flush=True)
        print(n_cams, "2L", res["two_level"][0], "%.6e %.4f" % res["two_level"][1:], flush=True)
