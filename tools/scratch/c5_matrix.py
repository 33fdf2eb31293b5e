import os, sys, subprocess, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.synthetic import make_bal_problem
    bal = make_bal_problem(1723, 156502, 678718, seed=0)
    mt, lag = float(sys.argv[1]), int(sys.argv[2])
    for label, kw in (("skip", dict(loss="huber", max_iters=50, ftol=1e-5, xtol=1e-5, gtol=1e-8, pcg_tol=0.1, pcg_max_iters=200)),
                      ("test_bal tolerances", dict(loss="huber", max_iters=60, ftol=1e-7, xtol=1e-10, gtol=1e-10, pcg_tol=0.1, pcg_max_iters=300))):
        if label == "skip":
            continue
        with hip_backend.Solver(0) as s:
            intr0 = s.set_problem_bal(bal, fixed_cam=0)
            for rep in range(2):
                s.set_params(bal.cams[:, :6], bal.pts)
                intr = intr0.copy()
                out = s.solve_bal_resident(intr, pcg_model_tol=mt, precond_lag=lag, **kw)
            tr = s.trace()
        print(f"floor {'off' if os.environ.get('BA_NO_CAP_FLOOR') else 'on '} model_tol {mt} lag {lag} [{label}]: {out['iterations']} LM, {out['pcg_iterations']} PCG, "
              f"RMSE {np.sqrt(out['final_sse'] / bal.n_obs):.6f}, cost {out['final_cost']:.6f}, {out['status_name']}, {out['seconds_total'] * 1e3:.1f} ms, PCG/LM {[t['pcg_iterations'] for t in tr]} lam {['%.0e' % t['damping'] for t in tr]} acc {''.join(str(int(t['accepted'])) for t in tr)}", flush=True)
else:
    for floor in ("on",):
        for mt in ("0", "0.5"):
            for lag in ("0", "3"):
                env = dict(os.environ)
                if floor == "off":
                    env["BA_NO_CAP_FLOOR"] = "1"
                subprocess.run([sys.executable, __file__, mt, lag], env=env)
