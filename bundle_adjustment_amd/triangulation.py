"""Two-view triangulation + cheirality and the re-observation bookkeeping of a new keyframe -- the step right upstream of
bundle adjustment that creates the landmarks it refines.

Mirrors ``VisualOdometryPipeline._triangulate_points`` (``src/pipeline.py:315-336``: same arguments, same return value
``(points_3d[:3, valid], valid_indices)`` or ``(None, None)``, same log line) on the GPU (``ba_triangulate``), and the
split of a keyframe's inlier matches into re-observations and new points (``src/pipeline.py:248-282``) as array code.
``cv2.triangulatePoints`` is restated from OpenCV's published DLT; the sign of its singular vector is fixed as
``w >= 0`` (parity unpinned at the cv2 boundary).  No CPU fallback: without the library / a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hip_backend


def triangulate_points(camera_matrix, R_rel, t_rel, pts1, pts2, solver=None, quiet=False):
    """-> (3 x n_kept points in the first camera's frame, indices kept), ``src/pipeline.py:315-336``."""
    pts1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    if pts1.shape[0] == 0:
        return None, None
    if pts2.shape[0] != pts1.shape[0]:
        raise ValueError("pts1 and pts2 must list the same number of points")
    own = solver is None
    s = hip_backend.Solver(0) if own else solver
    try:
        xyz, valid = s.triangulate(camera_matrix, R_rel, t_rel, pts1, pts2)
    finally:
        if own:
            s.close()
    if not quiet:
        print(f"    -> Triangulation: Kept {int(valid.sum())} of {pts1.shape[0]} points.")
    return xyz[valid].T.copy(), np.where(valid)[0]


def split_reobservations(last_kf_observations, query_idx, train_idx):
    """The bookkeeping loop of ``src/pipeline.py:251-282`` over the inlier matches of a new keyframe, as arrays.
    ``last_kf_observations``: the last keyframe's ``[(mp_id, kp_idx)]``; ``query_idx`` / ``train_idx``: keypoint index
    in the last / new keyframe per inlier match.  Returns ``(is_reobs, mp_id)``: for match i, ``is_reobs[i]`` says the
    last keyframe's keypoint already has a landmark (``mp_id[i]``; the LAST one listed for that keypoint, as the dict
    comprehension of :251 keeps), else it is a new point to triangulate (``mp_id[i] = -1``)."""
    q = np.asarray(query_idx, dtype=np.int64).ravel()
    if len(last_kf_observations) == 0 or q.size == 0:
        return np.zeros(q.size, dtype=bool), -np.ones(q.size, dtype=np.int64)
    ob = np.asarray(last_kf_observations, dtype=np.int64).reshape(-1, 2)
    # later entries overwrite earlier ones for the same keypoint index
    order = np.argsort(ob[:, 1], kind="stable")
    kp_sorted, mp_sorted = ob[order, 1], ob[order, 0]
    last_of_run = np.r_[kp_sorted[1:] != kp_sorted[:-1], True]
    kp_u, mp_u = kp_sorted[last_of_run], mp_sorted[last_of_run]
    pos = np.searchsorted(kp_u, q)
    pos_c = np.minimum(pos, kp_u.size - 1)
    hit = kp_u[pos_c] == q
    return hit, np.where(hit, mp_u[pos_c], -1)
