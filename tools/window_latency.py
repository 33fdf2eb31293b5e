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
# a map like the reference's pipeline grows: landmarks are triangulated from two consecutive keyframes and tracked over the
# next few (src/pipeline.py:248-308), so a 5-keyframe window sees most of its landmarks from several cameras
# (make_problem's uniformly random visibility would leave most of a window's landmarks with ONE observation inside it:
# free depths, dozens of LM iterations -- not what a sliding window is)
from bundle_adjustment_amd.synthetic import make_bal_like
pl = make_bal_like(n_cams=n_cams + extra, n_pts=n_pts * 6, n_obs_target=n_pts * 6 * k, seed=1)
# (the first pass also pays for device buffers growing as the windows change size: it is printed, but it is the warm-up)
for pass_no, reuse in enumerate((False, True, False)):
    full = problem_to_map(pl, extra_newest=False)       # a fresh (unoptimised) map for each mode
    gv = Map()
    gv.map_points = full.map_points
    ids = sorted(full.keyframes)
    for i in ids[:n_cams + 1]:
        gv.add_keyframe(full.keyframes[i])
    bs = BundleAdjuster(K, window_size=n_cams, reuse_window=reuse)
    lat = []
    # stage timers around the solver calls of run() (the rest of a call is host glue: window walk / cache, write-back, prints)
    sv = bs._get_solver()
    stage = {"set_problem": [], "set_params": [], "solve": [], "get_params": [], "get_rotations": []}
    iters = []
    def timed(name):
        fn = getattr(sv, name)
        def w(*a, **k):
            t0 = time.perf_counter(); r = fn(*a, **k); stage[name].append(time.perf_counter() - t0)
            if name == "solve": iters.append(r["iterations"])
            return r
        setattr(sv, name, w)
    for nm in stage: timed(nm)
    for i in ids[n_cams + 1:]:
        buf = io.StringIO()
        t = time.perf_counter()
        with redirect_stdout(buf):
            bs.run(gv)
        lat.append(time.perf_counter() - t)
        gv.add_keyframe(full.keyframes[i])
    lat = np.array(lat[3:]) * 1e3
    extra_note = f", cache hits {bs._window.hits}" if reuse else ""
    print(f"  sliding window{' (warm-up pass)' if pass_no == 0 else ''}, reuse_window={reuse}: run() median {np.median(lat):7.3f} ms   min {lat.min():7.3f}{extra_note}")
    print("     inside run(): " + ", ".join(f"{nm} {1e3 * np.median(v):.3f} ms x{len(v)}" for nm, v in stage.items() if v)
          + f"; LM iterations per solve: median {np.median(iters):.0f}, max {max(iters)}")
    bs.close()
