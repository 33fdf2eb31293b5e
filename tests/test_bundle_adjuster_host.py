"""CPU: host-side behaviour of the drop-in BundleAdjuster against the golden vectors
captured from the imported reference (window selection, gather order, packing, skip and
divergence paths, log lines, write-back).  The device is replaced by an oracle-backed test
double (tests/fake_solver.py); the GPU versions of these checks are in test_gpu_run.py."""
import io
import re
from contextlib import redirect_stdout

import numpy as np
import pytest

from bundle_adjustment_amd import bundle_adjuster as ba_mod
from bundle_adjustment_amd.problem import (flatten_map_window, flatten_map_window_numpy, flatten_window,
                                           gather_window)
from bundle_adjustment_amd.synthetic import make_problem, problem_to_map
from tests.fake_solver import OracleSolver
from tests.helpers import golden_flat_problem, load_golden, rebuild_map

LOG_RE = re.compile(r"^    -> LBA Complete\. Initial Cost: (\d+\.\d\d), Final Cost: (\d+\.\d\d), Improvement: (-?\d+\.\d\d)%$")


@pytest.fixture()
def fake_device(monkeypatch):
    monkeypatch.setattr(ba_mod.hip_backend, "Solver", OracleSolver)
    OracleSolver.force_diverge = False
    return OracleSolver


def _run(ba, gmap):
    buf = io.StringIO()
    with redirect_stdout(buf):
        ba.run(gmap)
    return buf.getvalue()


@pytest.mark.parametrize("name", ["cost_seed0", "cost_edge"])
def test_gather_and_flatten_match_reference_layout(name):
    """Same window, same observation (row) order, same pixels, same x0 as the reference."""
    g = load_golden(name)
    gmap = rebuild_map(g, prefix="map_")
    w = int(g["window_size"])
    local = sorted(gmap.keyframes)[-(w + 1):-1]
    assert local == [int(i) for i in g["local_kf_ids"]]
    mp_ids, observations, kp2d = gather_window(gmap, local)
    assert mp_ids == [int(i) for i in g["mp_ids"]]
    assert observations == [(int(a), int(b)) for a, b in g["observations"]]
    np.testing.assert_array_equal(np.array([kp2d[o] for o in observations]), g["uv_rows"])
    p = flatten_window(gmap, local, mp_ids, observations, kp2d, g["K"])
    q = golden_flat_problem(g)
    np.testing.assert_array_equal(p.cam_idx, q.cam_idx)
    np.testing.assert_array_equal(p.pt_idx, q.pt_idx)
    np.testing.assert_allclose(p.cams, q.cams, atol=1e-12)     # rvec of R (incl. the non-orthogonal one) and t
    np.testing.assert_array_equal(p.pts, q.pts)
    # the array-level pass used by run() builds the identical problem
    f, f_ids = flatten_map_window(gmap, local, g["K"])
    assert f_ids == mp_ids
    for a, b in ((f.cam_idx, p.cam_idx), (f.pt_idx, p.pt_idx), (f.uv, p.uv), (f.cams, p.cams), (f.pts, p.pts), (f.K4, p.K4)):
        np.testing.assert_array_equal(a, b)
    n, n_ids = flatten_map_window_numpy(gmap, local, g["K"])
    assert n_ids == mp_ids
    for a, b in ((n.cam_idx, p.cam_idx), (n.pt_idx, p.pt_idx), (n.uv, p.uv), (n.cams, p.cams), (n.pts, p.pts)):
        np.testing.assert_array_equal(a, b)
    na = len(local) - 1
    x0 = np.concatenate([p.cams[1:, :3].ravel(), p.cams[1:, 3:].ravel(), p.pts.ravel()])
    np.testing.assert_allclose(x0, g["x0"], atol=1e-12)
    assert x0.size == 6 * na + 3 * len(mp_ids)


def test_native_map_walk_matches_the_tuple_and_dict_walk():
    """csrc/mapwalk.c against gather_window + flatten_window on a map with everything the reference's
    walk tolerates: a map point listed twice in one keyframe (last keypoint wins for both rows), culled
    map points, list rows, a negative keypoint index, positions as (3,1), (3,) and plain lists, sparse and
    negative ids, an empty keyframe."""
    rng = np.random.default_rng(11)
    p = make_problem(6, 300, 4, seed=21)
    gmap = problem_to_map(p)
    ids = sorted(gmap.keyframes)
    kf = gmap.keyframes[ids[1]]
    first_mp, first_kp = kf.observations[0]
    kf.observations.append((first_mp, len(kf.keypoints) - 1))            # repeated pair: later keypoint wins
    kf.observations[3] = list(kf.observations[3])                         # a list row
    mp5, _ = kf.observations[5]
    kf.observations[5] = (mp5, -2)                                        # python-style negative index
    gmap.keyframes[ids[3]].observations = []                              # empty keyframe inside the window
    for m in rng.choice(sorted(gmap.map_points), 25, replace=False):      # culled points
        del gmap.map_points[int(m)]
    keys = sorted(gmap.map_points)
    gmap.map_points[keys[0]].position = gmap.map_points[keys[0]].position.reshape(3)
    gmap.map_points[keys[1]].position = [float(v) for v in gmap.map_points[keys[1]].position.ravel()]
    gmap.map_points[keys[2]].position = np.asfortranarray(gmap.map_points[keys[2]].position.reshape(1, 3)).T
    # sparse / negative ids: rename two map points everywhere
    for old, new in ((keys[3], 10**12 + 7), (keys[4], -5)):
        gmap.map_points[new] = gmap.map_points.pop(old)
        for k in ids:
            gmap.keyframes[k].observations = [r if r[0] != old else (new, r[1]) for r in gmap.keyframes[k].observations]
    K = np.array([[500.0, 0, 320], [0, 510.0, 240], [0, 0, 1]])
    for local in (ids[:-1], ids[2:5], [ids[3]]):
        mp_ids, observations, kp2d = gather_window(gmap, local)
        f, f_ids = flatten_map_window(gmap, local, K)
        n, n_ids = flatten_map_window_numpy(gmap, local, K)
        if not mp_ids:
            assert f is None and f_ids == [] and n is None and n_ids == []
            continue
        q = flatten_window(gmap, local, mp_ids, observations, kp2d, K)
        assert f_ids == mp_ids == n_ids
        for got in (f, n):
            assert got.cam_idx.dtype == np.int32 and got.pt_idx.dtype == np.int32
            for a, b in ((got.cam_idx, q.cam_idx), (got.pt_idx, q.pt_idx), (got.uv, q.uv), (got.cams, q.cams),
                         (got.pts, q.pts), (got.K4, q.K4)):
                np.testing.assert_array_equal(a, b)
    # errors surface as Python exceptions, not crashes
    gmap.keyframes[ids[1]].observations[-1] = (first_mp, 10**6)            # the winning (last) row of that pair
    with pytest.raises(IndexError):
        flatten_map_window(gmap, ids[:-1], K)
    with pytest.raises(KeyError):
        flatten_map_window(gmap, [12345], K)


@pytest.mark.parametrize("name", ["cost_seed1", "cost_edge"])
def test_sparsity_helper_matches_reference(name):
    g = load_golden(name)
    ba = ba_mod.BundleAdjuster(g["K"], window_size=int(g["window_size"]))
    obs = [(int(a), int(b)) for a, b in g["observations"]]
    A = ba._prepare_sparsity_matrix(len(g["adj_kf_ids"]), len(g["mp_ids"]), [int(i) for i in g["adj_kf_ids"]],
                                    [int(i) for i in g["mp_ids"]], obs).tocoo()
    order = np.lexsort((A.col, A.row))
    np.testing.assert_array_equal(A.row[order], g["sp_rows"])
    np.testing.assert_array_equal(A.col[order], g["sp_cols"])
    assert A.shape == tuple(g["sp_shape"])


def test_skip_paths_print_reference_lines(fake_device):
    g = load_golden("run_skips")
    K = np.eye(3)
    p = make_problem(3, 30, 2, seed=3)
    gmap = problem_to_map(p)                                   # 4 keyframes < default window 5
    ba = ba_mod.BundleAdjuster(K)
    assert ba.window_size == int(g["default_window"]) == 5
    assert _run(ba, gmap) == str(g["log_few"])
    p = make_problem(1, 30, 1, seed=4)
    assert _run(ba_mod.BundleAdjuster(K, window_size=1), problem_to_map(p)) == str(g["log_noadj"])
    p = make_problem(3, 30, 2, seed=5)
    gmap = problem_to_map(p)
    gmap.map_points.clear()
    assert _run(ba_mod.BundleAdjuster(K, window_size=3), gmap) == str(g["log_nopts"])


@pytest.mark.parametrize("name", ["run_seed0", "run_seed1", "run_global"])
def test_run_writes_back_like_reference(fake_device, name):
    g = load_golden(name)
    gmap = rebuild_map(g)
    ba = ba_mod.BundleAdjuster(g["K"], window_size=int(g["window_size"]))
    log = _run(ba, gmap)
    lines = log.splitlines()
    ref_lines = str(g["log"]).splitlines()
    assert lines[0] == ref_lines[0] == "    --- Running Local Bundle Adjustment ---"
    m, mref = LOG_RE.match(lines[1]), LOG_RE.match(ref_lines[1])
    assert m and mref, lines[1]
    assert m.group(1) == mref.group(1)                          # identical initial cost to the cent
    assert float(m.group(2)) <= float(mref.group(2)) * (1 + 1e-9)   # never worse than the reference's result
    # same keyframes / points touched, same shapes, untouched ones identical to before
    w = int(g["window_size"])
    ids = sorted(gmap.keyframes)
    local = ids[-(w + 1):-1]
    for n, i in enumerate(ids):
        kf = gmap.keyframes[i]
        assert kf.R.shape == (3, 3) and kf.t.shape == (3, 1)
        changed_ref = not np.array_equal(g["before_R"][n], g["after_R"][n])
        assert changed_ref == (i in local[1:])
        if i not in local[1:]:
            np.testing.assert_array_equal(kf.R, g["before_R"][n])
            np.testing.assert_array_equal(kf.t.ravel(), g["before_t"][n])
        else:
            np.testing.assert_allclose(kf.R @ kf.R.T, np.eye(3), atol=1e-12)
    for n, j in enumerate(sorted(gmap.map_points)):
        assert gmap.map_points[j].position.shape == (3, 1)
        if np.array_equal(g["before_X"][n], g["after_X"][n]):   # point outside the window in the reference run
            np.testing.assert_array_equal(gmap.map_points[j].position.ravel(), g["before_X"][n])


def test_divergence_guard_leaves_map_untouched(fake_device):
    g = load_golden("run_seed0")
    gmap = rebuild_map(g)
    fake_device.force_diverge = True
    log = _run(ba_mod.BundleAdjuster(g["K"], window_size=int(g["window_size"])), gmap)
    assert re.search(r"-> LBA Diverged! Cost increased from \d+\.\d\d to \d+\.\d\d\. Discarding results\.", log)
    for n, i in enumerate(sorted(gmap.keyframes)):
        np.testing.assert_array_equal(gmap.keyframes[i].R, g["before_R"][n])
        np.testing.assert_array_equal(gmap.keyframes[i].t.ravel(), g["before_t"][n])
    for n, j in enumerate(sorted(gmap.map_points)):
        np.testing.assert_array_equal(gmap.map_points[j].position.ravel(), g["before_X"][n])


def test_global_ba_window_idiom(fake_device):
    """main.py:83-86: window_size = number of keyframes -> every keyframe but the newest."""
    p = make_problem(5, 60, 3, seed=6)
    gmap = problem_to_map(p)                                    # 6 keyframes
    ba = ba_mod.BundleAdjuster(np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]]))
    original = ba.window_size
    ba.window_size = gmap.next_keyframe_id
    log = _run(ba, gmap)
    ba.window_size = original
    assert "LBA Complete" in log and ba._solver.n_cams == 5


def test_pcd_snapshot_only_when_directory_exists(fake_device, tmp_path, monkeypatch):
    p = make_problem(4, 40, 3, seed=7)
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    monkeypatch.setitem(ba_mod.DEBUG_DIRS, "lba_steps", str(tmp_path / "missing"))
    assert "Saved intermediate map" not in _run(ba_mod.BundleAdjuster(K, window_size=4), problem_to_map(p))
    d = tmp_path / "lba"
    d.mkdir()
    monkeypatch.setitem(ba_mod.DEBUG_DIRS, "lba_steps", str(d))
    log = _run(ba_mod.BundleAdjuster(K, window_size=4), problem_to_map(p))
    assert "Saved intermediate map to" in log
    txt = (d / "map_after_lba_kf_0.pcd").read_text()
    assert "POINTS 40" in txt and txt.count("\n") == 11 + 40


def test_metrics_sink_writes_one_json_line_per_run(fake_device, tmp_path):
    """SURVEY.md section 5 (metrics / logging): next to the reference's log line, run() can append one JSON object per
    call -- sizes, costs, RMSE, iteration counts and (from the device library) the per-iteration trace."""
    import json
    p = make_problem(6, 150, 4, seed=12)
    gmap = problem_to_map(p)
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    path = tmp_path / "lba_metrics.jsonl"
    ba = ba_mod.BundleAdjuster(K, window_size=p.n_cams, metrics_path=str(path))
    _run(ba, gmap)
    _run(ba, gmap)
    lines = [json.loads(ln) for ln in open(path)]
    assert len(lines) == 2
    m = lines[0]
    assert m["event"] == "lba" and m["n_keyframes"] == p.n_cams and m["n_landmarks"] == p.n_pts and m["n_observations"] == p.n_obs
    assert m["fixed_keyframe"] == 0 and m["final_sse"] < m["initial_sse"] and m["final_rmse_px"] < m["initial_rmse_px"]
    assert isinstance(m["trace"], list)
