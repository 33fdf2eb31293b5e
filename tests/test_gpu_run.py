"""GPU: the drop-in BundleAdjuster end to end on maps rebuilt from the reference's goldens."""
import io
import os
import re
from contextlib import redirect_stdout

import numpy as np
import pytest

from bundle_adjustment_amd import BundleAdjuster, hip_backend
from bundle_adjustment_amd.synthetic import make_config, problem_to_map
from oracle import ba_oracle as o
from tests.helpers import golden_cost_case, load_golden, rebuild_map

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG_RE = re.compile(r"^    -> LBA Complete\. Initial Cost: (\d+\.\d\d), Final Cost: (\d+\.\d\d), Improvement: (-?\d+\.\d\d)%$")


def _run(ba, gmap):
    buf = io.StringIO()
    with redirect_stdout(buf):
        ba.run(gmap)
    return buf.getvalue()


@pytest.mark.parametrize("name", ["cost_seed0", "cost_edge"])
def test_cost_function_entry_point(name):
    """BundleAdjuster._cost_function (same signature as the reference's) on the GPU ==
    the imported reference's output vector."""
    g = load_golden(name)
    ba = BundleAdjuster(g["K"], window_size=int(g["window_size"]))
    kw = golden_cost_case(g)
    for xk, fk in (("x0", "f0"), ("x1", "f1")):
        f = ba._cost_function(g[xk], **kw)
        assert f.shape == g[fk].shape
        assert np.abs(f - g[fk]).max() <= 1e-9 * max(1.0, np.abs(g[fk]).max())
    ba.close()


@pytest.mark.parametrize("name", ["run_seed0", "run_seed1", "run_global"])
def test_run_on_reference_maps(name):
    g = load_golden(name)
    gmap = rebuild_map(g)
    ba = BundleAdjuster(g["K"], window_size=int(g["window_size"]))
    lines = _run(ba, gmap).splitlines()
    ref = str(g["log"]).splitlines()
    assert lines[0] == ref[0]
    m, mref = LOG_RE.match(lines[1]), LOG_RE.match(ref[1])
    assert m and mref
    assert m.group(1) == mref.group(1)
    assert float(m.group(2)) <= float(mref.group(2)) * (1 + 1e-9)
    # the written-back map reproduces the reported final cost through the oracle residual
    w = int(g["window_size"])
    local = sorted(gmap.keyframes)[-(w + 1):-1]
    mp_ids, observations, kp2d = ba._gather_local_data(gmap, local)
    adj = local[1:]
    x = np.concatenate([np.array([o.rodrigues_to_vec(gmap.keyframes[i].R) for i in adj]).ravel(),
                        np.array([gmap.keyframes[i].t.ravel() for i in adj]).ravel(),
                        np.array([gmap.map_points[i].position.ravel() for i in mp_ids]).ravel()])
    f = o.reference_cost_function(x, (gmap.keyframes[local[0]].R, gmap.keyframes[local[0]].t), local[0], adj, mp_ids,
                                  observations, kp2d, g["K"])
    assert abs(float((f ** 2).sum()) - ba.last_summary["final_sse"]) <= 1e-6 * ba.last_summary["final_sse"]
    ba.close()


def test_run_on_synthetic_c2_map():
    p = make_config("C2", seed=2)
    gmap = problem_to_map(p)
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    ba = BundleAdjuster(K, window_size=p.n_cams)
    log = _run(ba, gmap)
    m = LOG_RE.match(log.splitlines()[1])
    assert m and float(m.group(3)) > 90.0
    ba.close()


def test_integration_md_binding_runs():
    """The ctypes binding printed in INTEGRATION.md (what a reference maintainer would paste) is
    executed as written -- only the library path and the cv2.Rodrigues call are bound to local names --
    and must reproduce the Solver class's result."""
    import re
    import types
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.rotations import matrices_to_rvecs, rvecs_to_matrices
    from bundle_adjustment_amd.synthetic import make_problem
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.findall(r"```python\n(.*?)```", text, re.S)[1]
    assert 'C.CDLL("libba_hip.so")' in code
    code = code.replace('C.CDLL("libba_hip.so")', f"C.CDLL({hip_backend.LIB_PATH!r})")
    cv2 = types.SimpleNamespace(Rodrigues=lambda R: (matrices_to_rvecs(np.asarray(R, dtype=np.float64)[None])[0].reshape(3, 1), None))
    ns = {"cv2": cv2}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    p = make_problem(6, 300, 4, seed=2)
    observations = [(int(c), int(m)) for c, m in zip(p.cam_idx, p.pt_idx)]
    kp = {o: tuple(uv) for o, uv in zip(observations, p.uv)}
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    R0 = rvecs_to_matrices(p.cams[:1, :3])[0]
    rv, tv, pts, summ = ns["solve_window"](K, R0, p.cams[0, 3:], p.cams[1:, :3], p.cams[1:, 3:], p.pts, observations, kp,
                                           {i: i for i in range(p.n_cams)}, {i: i for i in range(p.n_pts)})
    # The binding turns the fixed keyframe's R back into a rotation vector (as :59 does); give the Solver class the very
    # same numbers and the two must agree BIT FOR BIT: same library, same inputs, every sum in a fixed order.
    q = type(p)(p.cams.copy(), p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0)
    q.cams[0, :3] = matrices_to_rvecs(R0[None])[0]
    with hip_backend.Solver(0) as s:
        s.set_problem(q)
        ref = s.solve()
        cams_ref, pts_ref = s.get_params()
    assert summ.iterations == ref["iterations"] and summ.accepted == ref["accepted"] and summ.pcg_iterations == ref["pcg_iterations"]
    assert summ.initial_sse == ref["initial_sse"] and summ.final_sse == ref["final_sse"] and summ.final_cost == ref["final_cost"]
    assert summ.final_sse < 0.05 * summ.initial_sse
    np.testing.assert_array_equal(rv, cams_ref[1:, :3])
    np.testing.assert_array_equal(tv, cams_ref[1:, 3:])
    np.testing.assert_array_equal(pts, pts_ref)


def test_iteration_trace_matches_the_summary():
    """ba_get_trace: one record per LM iteration of the last solve, consistent with the summary."""
    from bundle_adjustment_amd.synthetic import make_problem
    p = make_problem(10, 800, 5, seed=31, outlier_frac=0.02)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        assert s.trace() == []
        out = s.solve(loss="huber", max_iters=25, ftol=1e-10, xtol=1e-12, gtol=1e-12, pcg_tol=1e-2)
        tr = s.trace()
    assert len(tr) == out["iterations"] and [t["iteration"] for t in tr] == list(range(1, len(tr) + 1))
    assert sum(t["pcg_iterations"] for t in tr) == out["pcg_iterations"]
    assert sum(t["accepted"] for t in tr) == out["accepted"]
    assert abs(tr[0]["cost"] - out["initial_cost"]) <= 1e-12 * out["initial_cost"]
    last_acc = [t for t in tr if t["accepted"]][-1]
    assert abs(last_acc["cost_trial"] - out["final_cost"]) <= 1e-12 * out["final_cost"]
    for t in tr:
        assert (t["cost_trial"] < t["cost"]) == t["accepted"] or not np.isfinite(t["cost_trial"])
        assert t["damping"] > 0 and t["seconds"] > 0 and t["step_norm"] >= 0
