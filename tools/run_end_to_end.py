"""End-to-end BundleAdjuster.run on a synthetic Map: host flattening (csrc/mapwalk.c) + upload +
GPU solve + write-back, with the split of where the wall time goes."""
import io, os, sys, time
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import BundleAdjuster, hip_backend
from bundle_adjustment_amd.problem import flatten_map_window, flatten_map_window_numpy
from bundle_adjustment_amd.synthetic import make_config, problem_to_map

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
p = make_config(cfg)
t = time.time(); g = problem_to_map(p); print(f"{cfg}: map build {time.time()-t:.2f}s ({p.n_cams} keyframes, {p.n_pts} points, {p.n_obs} observations)")
K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
ids = sorted(g.keyframes)
for name, f in (("native walk", flatten_map_window), ("numpy walk", flatten_map_window_numpy)):
    t = time.time(); prob, _ = f(g, ids, K); print(f"  flatten ({name}): {1e3*(time.time()-t):.1f} ms")
s = hip_backend.Solver(0)
t = time.time(); s.set_problem(prob); s.synchronize(); print(f"  set_problem (sorts + upload): {1e3*(time.time()-t):.1f} ms")
t = time.time(); s.set_problem(prob); s.synchronize(); print(f"  set_problem again: {1e3*(time.time()-t):.1f} ms")
s.close()
ba = BundleAdjuster(K, window_size=p.n_cams)
for rep in range(3):      # 0: cold (walk + sorts + upload); 1, 2: the window is unchanged -> cached structure, parameters only
    t = time.time()
    buf = io.StringIO()
    with redirect_stdout(buf):
        ba.run(g)
    dt = time.time() - t
    print(f"  run #{rep}: {1e3*dt:.1f} ms total, solve {ba.last_summary['seconds_total']*1e3:.1f} ms "
          f"({ba.last_summary['iterations']} LM it) | {buf.getvalue().strip().splitlines()[-1].strip()} | cache {ba._window.hits}")
x = np.concatenate([prob.cams[1:, :3].ravel(), prob.cams[1:, 3:].ravel(), prob.pts.ravel()])     # prob: every keyframe of the map
mp_ids = np.array(sorted(g.map_points), dtype=np.int64)
t = time.time(); ba._update_map(g, x, ids[1:], mp_ids); print(f"  _update_map alone (native rebind): {1e3*(time.time()-t):.1f} ms")
