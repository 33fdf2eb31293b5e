#!/usr/bin/env python3
"""GPU box: drive the device solver to convergence on C2 / C3 (seed 0) with the Huber loss and export the minimiser x*
in the reference's parameter packing [rvec(Na,3) | tvec(Na,3) | points(Np,3)] (src/bundle_adjuster.py:157-162), for
tests/golden/make_golden_converged.py --certify (build container: evaluates the imported reference's _cost_function and the
Huber gradient at x*, and restarts scipy from it).

    python tools/export_converged.py [C2 C3 ...]  ->  gpurun_out/xstar_<cfg>_huber.npy + .json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bundle_adjustment_amd import hip_backend                              # noqa: E402
from bundle_adjustment_amd.synthetic import make_config                    # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
for cfg in (sys.argv[1:] or ["C2", "C3"]):
    p = make_config(cfg, seed=0)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        # two stages: the descent of the GPU test (PCG tolerance 1e-2), then a polish with a tight inner solve
        out = s.solve(loss="huber", max_iters=150, ftol=1e-14, xtol=1e-14, gtol=0.0, pcg_tol=1e-2, pcg_max_iters=500)
        pol = s.solve(loss="huber", max_iters=60, ftol=1e-16, xtol=1e-16, gtol=0.0, pcg_tol=1e-6, pcg_max_iters=2000, pcg_model_tol=0.0)
        cams, pts = s.get_params()
    adj = [i for i in range(p.n_cams) if i != p.fixed_cam]
    x = np.concatenate([cams[adj, :3].ravel(), cams[adj, 3:].ravel(), pts.ravel()])
    np.save(os.path.join(out_dir, f"xstar_{cfg.lower()}_huber.npy"), x)
    info = dict(config=cfg, seed=0, n=int(x.size), descent=out, polish=pol,
                rmse_descent=float(np.sqrt(out["final_sse"] / p.n_obs)), rmse_polish=float(np.sqrt(pol["final_sse"] / p.n_obs)))
    with open(os.path.join(out_dir, f"xstar_{cfg.lower()}_huber.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(cfg, "descent:", out["iterations"], "LM it, cost", repr(out["final_cost"]), "| polish:", pol["iterations"], "LM it, cost",
          repr(pol["final_cost"]), "rmse", info["rmse_polish"], flush=True)
