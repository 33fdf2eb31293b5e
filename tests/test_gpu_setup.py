"""GPU: ba_set_problem's DEVICE build of the problem layout (csrc/ba_setup.hpp: histogram / scan / scatter / segment sorts
on the GPU from one upload of the caller's arrays) against the host build it replaces for large problems (BA_SETUP=host:
the counting sorts of csrc/ba_hip.hip) -- every array and every grid scalar, bit for bit -- and a solve through either."""
import numpy as np
import pytest

from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.problem import BAProblem
from bundle_adjustment_amd.synthetic import make_config, make_problem

pytestmark = pytest.mark.gpu
ARRAYS = ("pt_off", "p_cam", "c_pt", "c_orig", "offk", "long_pts", "blk_win", "slot", "p_uv", "c_uv")


def _layout(p, mode, monkeypatch):
    monkeypatch.setenv("BA_SETUP", mode)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        lay = {k: s.debug_layout(k) for k in ARRAYS}
        sc = s.debug_layout("scalars")
        out = s.solve(loss="huber", max_iters=4, ftol=0.0, xtol=0.0, gtol=0.0)
        par = s.get_params()
    return lay, sc, out, par


def _long_track_problem(seed):
    """random visibility plus a few hundred landmarks seen by MANY cameras (tracks of 40 .. 300 observations: the 16-lane
    rows of the point passes, the workgroup rank sort of the device build), duplicated (camera, landmark) pairs, landmarks
    without any observation, in shuffled observation order"""
    rng = np.random.default_rng(seed)
    # (600 cameras: one point-pass workgroup per compute unit; 70 000 landmarks: two lanes per landmark, so long tracks get rows)
    p = make_problem(600, 70000, 4, seed=seed, outlier_frac=0.02)
    extra_c, extra_p = [], []
    for q in rng.choice(p.n_pts, 300, replace=False):
        n = int(rng.integers(40, 300))
        extra_c.append(rng.integers(0, p.n_cams, n)); extra_p.append(np.full(n, q))
    ci = np.concatenate([p.cam_idx] + extra_c).astype(np.int32)
    pi = np.concatenate([p.pt_idx] + extra_p).astype(np.int32)
    keep = pi % 97 != 5                                           # landmarks nobody sees
    ci, pi = ci[keep], pi[keep]
    uv = np.concatenate([p.uv, rng.uniform(0, 700, (ci.size, 2)).astype(np.float32).astype(np.float64)])[:ci.size]
    perm = rng.permutation(ci.size)
    return BAProblem(p.cams, p.pts, ci[perm], pi[perm], np.ascontiguousarray(uv[perm]), p.K4, 0)


@pytest.mark.parametrize("case", ["C2x8", "long_tracks", "C3"])
def test_device_build_equals_the_host_build(case, monkeypatch):
    if case == "C3":
        p = make_config("C3", seed=0)
    elif case == "C2x8":
        p = make_problem(400, 40000, 6, seed=5, outlier_frac=0.01)
    else:
        p = _long_track_problem(9)
    dev, sc_d, out_d, par_d = _layout(p, "device", monkeypatch)
    host, sc_h, out_h, par_h = _layout(p, "host", monkeypatch)
    assert sc_d["build_path"] == 1 and sc_h["build_path"] == 0
    for k in sc_h:
        if k != "build_path":
            assert sc_d[k] == sc_h[k], (k, sc_d[k], sc_h[k])
    for k in ARRAYS:
        assert dev[k].shape == host[k].shape and np.array_equal(dev[k], host[k]), k
    if case == "long_tracks":
        assert sc_d["n_long"] > 100
    # the same layout: the same bits out of the solver
    assert out_d["final_cost"] == out_h["final_cost"] and out_d["pcg_iterations"] == out_h["pcg_iterations"]
    assert np.array_equal(par_d[0], par_h[0]) and np.array_equal(par_d[1], par_h[1])


def test_device_build_reports_a_bad_index_like_the_host_build(monkeypatch):
    monkeypatch.setenv("BA_SETUP", "device")
    p = make_problem(100, 8000, 5, seed=2)
    bad = p.pt_idx.copy()
    bad[1234] = p.n_pts
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        rc = s._lib.ba_set_problem(s._h, p.n_cams, p.n_pts, p.n_obs, p.cam_idx.ctypes.data_as(hip_backend._IP),
                                   bad.ctypes.data_as(hip_backend._IP), hip_backend._dp(p.uv), hip_backend._dp(p.K4), 0)
        assert rc == -1 and b"pt_idx[1234]" in s._lib.ba_last_error()
        out = s.solve(max_iters=3)                            # rejected before anything was touched: the old problem stands
        assert out["iterations"] == 3


@pytest.mark.parametrize("build", ["host", "device"])
def test_float32_valued_pixels_are_stored_as_float2_and_change_no_bit(build, monkeypatch):
    """cv2 keypoints are float32 values (/root/reference/src/bundle_adjuster.py:216): when every pixel of a problem is one,
    the multi-kernel path keeps its two pixel streams as float2 and widens on load (UvArr, csrc/ba_kernels.hpp) -- the same
    doubles reach the arithmetic, so every bit of the result is the one of the double2 streams (BA_PIXELS=f64).  One pixel
    that is NOT a float32 value keeps the whole problem on double2."""
    monkeypatch.setenv("BA_SETUP", build)
    p = make_problem(400, 40000, 6, seed=5, outlier_frac=0.01)
    assert np.array_equal(p.uv, p.uv.astype(np.float32).astype(np.float64))

    def run(prob, pixels):
        if pixels: monkeypatch.setenv("BA_PIXELS", pixels)
        else: monkeypatch.delenv("BA_PIXELS", raising=False)
        with hip_backend.Solver(0) as s:
            s.set_problem(prob)
            f32 = s.stats()["pixels_f32"]
            lay = {k: s.debug_layout(k) for k in ("p_uv", "c_uv")}
            out = s.solve(loss="huber", max_iters=5, ftol=0.0, xtol=0.0, gtol=0.0)
            cams, pts = s.get_params()
            r = s.residuals()[0]
        return f32, lay, out, cams, pts, r

    f_a, lay_a, out_a, cams_a, pts_a, r_a = run(p, None)
    f_b, lay_b, out_b, cams_b, pts_b, r_b = run(p, "f64")
    assert f_a == 1 and f_b == 0
    for k in lay_a:
        assert np.array_equal(lay_a[k], lay_b[k]), k
    assert out_a["final_cost"] == out_b["final_cost"] and out_a["pcg_iterations"] == out_b["pcg_iterations"]
    assert np.array_equal(cams_a, cams_b) and np.array_equal(pts_a, pts_b) and np.array_equal(r_a, r_b)
    # one pixel between two float32 values
    uv = p.uv.copy()
    uv[777, 1] += 1e-9
    q = BAProblem(p.cams, p.pts, p.cam_idx, p.pt_idx, uv, p.K4, 0)
    f_c, lay_c, out_c, *_ = run(q, None)
    assert f_c == 0 and out_c["status"] == 0
    src = np.flatnonzero(np.all(lay_c["p_uv"].reshape(-1, 2) == uv[777], axis=1))
    assert src.size >= 1                                      # (the odd pixel arrived intact)


def test_window_sized_problems_keep_double_pixels():
    p = make_config("C1", seed=0)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        assert s.stats()["pixels_f32"] == 0
