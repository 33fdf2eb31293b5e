"""Single-GPU timing of the problems one rank of an N-GPU C3 job sees (1000 cameras, 100k/N points),
with 2 lanes per point and with the lane count ba_set_problem picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend as hb
from bundle_adjustment_amd.synthetic import make_problem
for npts in (50000, 25000, 12500):
    p = make_problem(1000, npts, 10, seed=0)
    for lanes in ("2", "auto"):
        if lanes == "auto":
            os.environ.pop("BA_PT_LANES", None)
        else:
            os.environ["BA_PT_LANES"] = lanes
        s = hb.Solver(0)
        s.set_problem(p)
        kw = dict(loss="huber", max_iters=20, ftol=0, xtol=0, gtol=0)
        s.solve(**dict(kw, max_iters=3))
        s.set_params(p.cams, p.pts)
        out = s.solve(**kw)
        s.set_params(p.cams, p.pts)
        s.profile(reset=True)
        s.solve(profile=1, **kw)
        pr = s.profile()
        rate = out["iterations"] / out["seconds_total"]
        print("1000 cams / %6d pts, lanes %4s: %8.1f LM it/s   schur_pt %.2f us  linearize_pt %.2f us  backsub %.2f us"
              % (npts, lanes, rate, pr["schur_pt"]["working_mean_us"], pr["linearize_pt"]["working_mean_us"],
                 pr["backsub_pt"]["working_mean_us"]))
        s.close()
