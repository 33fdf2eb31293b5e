"""LM iterations/s and time per PCG iteration of a BAL-like chain problem of a given size (reference pinhole):
    python tools/bal_like_times.py N_CAMS N_PTS N_OBS      # BA_TIME_SETUP=1 also prints the point passes' grid"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import hip_backend as hb
from bundle_adjustment_amd.synthetic import make_bal_like
nc, npt, no = (int(x) for x in sys.argv[1:4])
p = make_bal_like(nc, npt, no, seed=0)
s = hb.Solver(0)
s.set_problem(p)
kw = dict(loss="huber", max_iters=20, ftol=0.0, xtol=0.0, gtol=1e-300, pcg_tol=0.1, pcg_max_iters=200)
ts = []
for r in range(6):
    s.set_params(p.cams, p.pts)
    t = time.perf_counter(); out = s.solve(**kw); ts.append(time.perf_counter() - t)
it = out["iterations"]; pcg = out["pcg_iterations"]
t = np.median(ts[1:])
print(f"{nc} cams {npt} pts {len(p.pt_idx)} obs: {it / t:8.1f} LM it/s, {pcg / it:6.1f} PCG/LM, {t / pcg * 1e6:7.2f} us per PCG iteration (all-in)")
