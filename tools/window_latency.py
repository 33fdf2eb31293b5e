"""Latency of one BundleAdjuster.run on a sliding-window-sized map (the reference's default use:
window_size 5, a few hundred landmarks), split by stage."""
import io, os, sys, time
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import BundleAdjuster, hip_backend
from bundle_adjustment_amd.problem import flatten_map_window
from bundle_adjustment_amd.synthetic import make_problem, problem_to_map

n_cams, n_pts, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 500, 4)
p = make_problem(n_cams, n_pts, k, seed=0)
K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
ba = BundleAdjuster(K, window_size=n_cams)
s = ba._get_solver()
times = {"flatten": [], "set_problem": [], "solve": [], "get": [], "run": []}
for rep in range(30):
    g = problem_to_map(p)                       # fresh (unoptimised) map every time
    ids = sorted(g.keyframes)[-(n_cams + 1):-1]
    t = time.perf_counter(); prob, _ = flatten_map_window(g, ids, K); times["flatten"].append(time.perf_counter() - t)
    t = time.perf_counter(); s.set_problem(prob); times["set_problem"].append(time.perf_counter() - t)
    t = time.perf_counter(); out = s.solve(**ba.solver_options); times["solve"].append(time.perf_counter() - t)
    t = time.perf_counter(); s.get_params(); s.get_rotations(); times["get"].append(time.perf_counter() - t)
    buf = io.StringIO()
    t = time.perf_counter()
    with redirect_stdout(buf):
        ba.run(g)
    times["run"].append(time.perf_counter() - t)
print(f"{n_cams} cams / {n_pts} pts / {p.n_obs} obs; {out['iterations']} LM iterations, {out['pcg_iterations']} PCG iterations")
for k_, v in times.items():
    v = np.array(v[5:]) * 1e3
    print(f"  {k_:12s} median {np.median(v):7.3f} ms   min {v.min():7.3f}")

# ---- the reference's actual use: run() after every new keyframe (src/pipeline.py:99), window sliding by one ----
from bundle_adjustment_amd.map_structures import Map
extra = 25
pl = make_problem(n_cams + extra, n_pts * 3, k, seed=1)
for reuse in (True, False):
    full = problem_to_map(pl, extra_newest=False)       # a fresh (unoptimised) map for each mode
    gv = Map()
    gv.map_points = full.map_points
    ids = sorted(full.keyframes)
    for i in ids[:n_cams + 1]:
        gv.add_keyframe(full.keyframes[i])
    bs = BundleAdjuster(K, window_size=n_cams, reuse_window=reuse)
    lat = []
    for i in ids[n_cams + 1:]:
        buf = io.StringIO()
        t = time.perf_counter()
        with redirect_stdout(buf):
            bs.run(gv)
        lat.append(time.perf_counter() - t)
        gv.add_keyframe(full.keyframes[i])
    lat = np.array(lat[3:]) * 1e3
    extra_note = f", cache hits {bs._window.hits}" if reuse else ""
    print(f"  sliding window, reuse_window={reuse}: run() median {np.median(lat):7.3f} ms   min {lat.min():7.3f}{extra_note}")
    bs.close()
