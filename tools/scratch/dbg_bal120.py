import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from bundle_adjustment_amd import hip_backend
    from tests.test_bal import _synthetic_bal
    p = _synthetic_bal(120, 6000, 6, seed=10)
    mt, lag = float(sys.argv[1]), int(sys.argv[2])
    with hip_backend.Solver(0) as s:
        out, cams, pts = s.solve_bal(p, loss="huber", max_iters=60, ftol=1e-8, xtol=1e-10, gtol=1e-10, pcg_tol=1e-1, pcg_max_iters=300, pcg_model_tol=mt, precond_lag=lag)
        tr = s.trace()
        st = s.stats()
    print(f"floor {'off' if os.environ.get('BA_NO_CAP_FLOOR') else 'on '} model {mt} lag {lag}: {out['iterations']} LM {out['pcg_iterations']} PCG {out['status_name']} cost {out['final_cost']:.9f} banded {st['banded']} raises {st['cap_floor_raises']}")
    print("   pcg", [t['pcg_iterations'] for t in tr])
    print("   lam", ["%.1e" % t['damping'] for t in tr])
    print("   acc", "".join("1" if t['accepted'] else "0" for t in tr))
else:
    for floor in ("off", "on"):
        for mt in ("0", "-1"):
            env = dict(os.environ)
            if floor == "off":
                env["BA_NO_CAP_FLOOR"] = "1"
            subprocess.run([sys.executable, __file__, mt, "3"], env=env)
