import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.synthetic import make_problem
    from oracle import ba_oracle as o
    loss = "huber"
    p = make_problem(12, 800, 5, seed=4, outlier_frac=0.02)
    kw = dict(max_iters=40, ftol=1e-13, xtol=1e-13, gtol=0.0, pcg_tol=1e-4, pcg_max_iters=400)
    lag = int(sys.argv[2])
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        out = s.solve(loss=loss, preconditioner="schur_jacobi", precond_lag=lag, **kw)
        tr = s.trace()
        cams, pts = s.get_params()
    ref = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, loss, precond="schur_jacobi", precond_lag=lag, **kw)
    print("riders", os.environ.get("BA_RIDERS"), "lag", lag, "iters", out["iterations"], ref["iterations"], "pcg", out["pcg_iterations"], ref["pcg_iters"],
          "cost rel", abs(out["final_cost"] - ref["cost"]) / ref["cost"], "cams", np.abs(cams - ref["cams"]).max(), "pts", np.abs(pts - ref["pts"]).max())
    for t, h in zip(tr, ref["history"]):
        print("  it %2d pcg %3d/%3d acc %d cost_trial %.15e / %.15e lam %.3e/%.3e" % (t["iteration"], t["pcg_iterations"], h["pcg"], t["accepted"], t["cost_trial"], h["cost_new"], t["damping"], h["lam"]))
else:
    for r in ("3", "7"):
        for lag in ("0", "3"):
            subprocess.run([sys.executable, __file__, "x", lag], env=dict(os.environ, BA_RIDERS=r))
