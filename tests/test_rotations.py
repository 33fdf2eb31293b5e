"""Host-side rotation matrix -> vector conversion (the reference's cv2.Rodrigues(matrix) call when a window is packed,
src/bundle_adjuster.py:157): the native path of the walk extension (polar projection by Newton's iteration) against the
numpy SVD route and the oracle's rodrigues_to_vec, incl. theta = 0, theta near pi, matrices that are not quite rotations,
and a singular matrix (falls back to the SVD route)."""
import numpy as np

from bundle_adjustment_amd import _mapwalk, rotations
from oracle import ba_oracle as o


def _svd_route(Rs):
    U, _, Vt = np.linalg.svd(Rs)
    return np.array([o.rodrigues_to_vec(q) for q in U @ Vt])


def test_native_conversion_matches_the_svd_route():
    rng = np.random.default_rng(0)
    near_pi = rng.normal(size=(20, 3))
    near_pi *= ((np.pi - 1e-6) / np.linalg.norm(near_pi, axis=1))[:, None]
    vecs = np.concatenate([rng.normal(size=(200, 3)), rng.normal(size=(40, 3)) * 1e-7, np.zeros((1, 3)),
                           np.array([[np.pi, 0, 0], [0, np.pi - 1e-9, 0], [0, 0, np.pi - 1e-7]]), near_pi])
    Rs = rotations.rvecs_to_matrices(vecs)
    for M, tol in ((Rs, 1e-14), (Rs + rng.normal(size=Rs.shape) * 1e-3, 1e-11)):
        M = np.ascontiguousarray(M)
        out = np.empty((M.shape[0], 3))
        assert _mapwalk.rvecs_from_matrices(M, out) == M.shape[0]
        ref = _svd_route(M)
        away = np.linalg.norm(ref, axis=1) < 3.0                      # (near pi the vector itself is only sqrt(eps)-accurate)
        assert np.abs(out - ref)[away].max() <= tol
        assert np.abs(rotations.rvecs_to_matrices(out) - rotations.rvecs_to_matrices(ref)).max() <= 1e-8
    # what matrices_to_rvecs returns for a window's worth is the native result; for many matrices the numpy one
    five = np.empty((5, 3)); _mapwalk.rvecs_from_matrices(np.ascontiguousarray(Rs[:5]), five)
    assert np.array_equal(rotations.matrices_to_rvecs(Rs[:5]), five)
    assert np.abs(rotations.matrices_to_rvecs(Rs) - _svd_route(Rs))[np.linalg.norm(vecs, axis=1) < 3.0].max() <= 1e-14


def test_singular_matrix_takes_the_svd_route():
    S = np.zeros((1, 3, 3)); S[0, 0, 0] = 1.0; S[0, 1, 1] = 1.0      # rank 2
    out = np.empty((1, 3))
    assert _mapwalk.rvecs_from_matrices(S, out) == -1
    v = rotations.matrices_to_rvecs(S)                              # numpy route: some rotation, finite
    assert v.shape == (1, 3) and np.isfinite(v).all()
