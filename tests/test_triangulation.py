"""Row f3: two-view triangulation + cheirality and the re-observation split.
CPU: the oracle's DLT recovers exact two-view geometry; the array form of the bookkeeping equals the reference's
dict / loop form (src/pipeline.py:251-282).  GPU: ba_triangulate against the oracle, <= 1e-9 relative on the points,
identical cheirality masks."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest

from bundle_adjustment_amd.rotations import rvecs_to_matrices
from bundle_adjustment_amd.triangulation import split_reobservations, triangulate_points
from oracle import ba_oracle as o

K = np.array([[912.7820434570312, 0.0, 650.2929077148438], [0.0, 913.0294189453125, 362.7241516113281], [0.0, 0.0, 1.0]])


def _scene(n, seed, noise=0.0, behind=0, baseline=0.4, rot=1.0):
    rng = np.random.default_rng(seed)
    R = rvecs_to_matrices(rot * np.array([[0.02, -0.05, 0.01]]))[0]
    t = baseline * np.array([[-1.0], [0.075], [0.125]])
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(5, 20, n)], axis=1)
    X[:behind, 2] *= -1.0                                        # points behind both cameras
    x1 = X @ K.T
    x2 = (X @ R.T + t.ravel()) @ K.T
    p1 = x1[:, :2] / x1[:, 2:] + rng.normal(0, noise, (n, 2))
    p2 = x2[:, :2] / x2[:, 2:] + rng.normal(0, noise, (n, 2))
    return R, t, X, p1, p2


def test_oracle_dlt_recovers_exact_geometry_and_cheirality():
    R, t, X, p1, p2 = _scene(200, 0, behind=7)
    xyz, valid = o.triangulate_points(K, R, t, p1, p2)
    np.testing.assert_allclose(xyz, X, rtol=2e-5, atol=2e-5)       # (the reference's + 1e-6 on w is a ~1e-5 relative bias)
    assert not valid[:7].any() and valid[7:].all()


def test_split_reobservations_equals_the_reference_loop():
    rng = np.random.default_rng(1)
    last_obs = [(int(rng.integers(0, 500)), int(k)) for k in rng.integers(0, 300, size=400)]      # duplicated keypoint indices too
    q = rng.integers(0, 350, size=500)
    tr = rng.integers(0, 1000, size=500)
    lookup = {kp: mp for mp, kp in last_obs}                     # src/pipeline.py:251
    want_hit = np.array([int(k) in lookup for k in q])
    want_mp = np.array([lookup.get(int(k), -1) for k in q])
    hit, mp = split_reobservations(last_obs, q, tr)
    np.testing.assert_array_equal(hit, want_hit)
    np.testing.assert_array_equal(mp, want_mp)
    hit0, mp0 = split_reobservations([], q, tr)
    assert not hit0.any() and (mp0 == -1).all()


@pytest.mark.gpu
def test_device_triangulation_matches_the_oracle():
    from bundle_adjustment_amd import hip_backend
    with hip_backend.Solver(0) as s:
        for n, seed, noise, behind in ((1, 3, 0.0, 0), (257, 4, 0.5, 11), (4000, 5, 1.0, 100)):
            R, t, X, p1, p2 = _scene(n, seed, noise, behind)
            xyz, valid = s.triangulate(K, R, t, p1, p2)
            ref, vref = o.triangulate_points(K, R, t, p1, p2)
            np.testing.assert_array_equal(valid, vref)
            assert np.abs(xyz - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        # low-parallax pairs (a keyframe right after the last one): baselines of 1 cm, 1 mm, 0.1 mm.  The DLT system is
        # then nearly rank-2; the device works on A itself (one-sided Jacobi SVD), not on A^T A, and stays on the
        # oracle's LAPACK SVD -- also in WHICH points it keeps (z > 0 in both cameras, src/pipeline.py:328-334)
        for base in (1e-2, 1e-3, 1e-4):
            Rb, tb, Xb, q1, q2 = _scene(3000, 6, 0.3, 40, baseline=base, rot=0.1)
            xyz_b, valid_b = s.triangulate(K, Rb, tb, q1, q2)
            ref_b, vref_b = o.triangulate_points(K, Rb, tb, q1, q2)
            scale = max(1.0, np.abs(ref_b).max())
            assert np.abs(xyz_b - ref_b).max() <= 1e-8 * scale, base
            z2 = (ref_b @ Rb.T + tb.ravel())[:, 2]
            decided = (np.abs(ref_b[:, 2]) > 1e-7 * scale) & (np.abs(z2) > 1e-7 * scale)      # not on the z = 0 knife edge
            np.testing.assert_array_equal(valid_b[decided], vref_b[decided])
            assert decided.mean() > 0.99
        # the reference-shaped wrapper: (3, kept) array, kept indices, its log line; empty input -> (None, None)
        buf = io.StringIO()
        with redirect_stdout(buf):
            pts, idx = triangulate_points(K, R, t.ravel(), p1, p2, solver=s)
        assert pts.shape == (3, int(vref.sum())) and np.array_equal(idx, np.where(vref)[0])
        assert buf.getvalue() == f"    -> Triangulation: Kept {int(vref.sum())} of {n} points.\n"
        np.testing.assert_allclose(pts.T, ref[vref], rtol=0, atol=1e-9 * np.abs(ref).max())
        assert triangulate_points(K, R, t, np.zeros((0, 2)), np.zeros((0, 2)), solver=s) == (None, None)


# ---- pinned by the reference's own function (tests/golden/make_golden.py section E) ------------------------------------------
SCENES = ("wide", "behind", "degenerate")


def _tri_golden():
    from tests.helpers import load_golden
    return load_golden("tri_scenes")


@pytest.mark.parametrize("name", SCENES)
def test_oracle_triangulation_matches_the_reference_function(name):
    """tri_scenes.npz holds what the imported, unmodified VisualOdometryPipeline._triangulate_points returned
    (src/pipeline.py:315-336) for three scenes: a wide baseline, one with points behind both cameras and points in front of
    the first but behind the second camera (each cheirality mask decides alone), and one with points at infinity where the
    '+ 1e-6' on w is what is divided by.  (cv2.triangulatePoints itself: the generator's numpy DLT -- parity unpinned at
    the cv2 boundary.)"""
    g = _tri_golden()
    xyz, valid = o.triangulate_points(g["K"], g[name + "_R"], g[name + "_t"], g[name + "_pts1"], g[name + "_pts2"])
    assert np.array_equal(np.where(valid)[0], g[name + "_idx"])
    want = g[name + "_out"]
    assert want.shape == (3, int(valid.sum()))
    assert np.abs(xyz[valid].T - want).max() <= 1e-10 * np.abs(want).max()
    assert str(g[name + "_log"]) == f"    -> Triangulation: Kept {int(valid.sum())} of {len(valid)} points.\n"


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_device_triangulation_matches_the_reference_function(name):
    """The drop-in wrapper (same arguments, return value and log line as the reference's method) on the device against
    the reference's recorded outputs: kept indices identical, points <= 1e-9 relative, the log line byte for byte."""
    g = _tri_golden()
    buf = io.StringIO()
    with redirect_stdout(buf):
        pts, idx = triangulate_points(g["K"], g[name + "_R"], g[name + "_t"], g[name + "_pts1"], g[name + "_pts2"])
    want = g[name + "_out"]
    assert np.array_equal(idx, g[name + "_idx"])
    # "degenerate": parallel rays leave the homogeneous w (~1e-8 of a unit vector) determined to ~1e-14 only, and the
    # result is x / (w + 1e-6): two correct SVDs (LAPACK's in the generator, the device's one-sided Jacobi) agree to 1e-7
    tol = 1e-9 if name != "degenerate" else 1e-7
    assert pts.shape == want.shape and np.abs(pts - want).max() <= tol * np.abs(want).max()
    assert buf.getvalue() == str(g[name + "_log"])
