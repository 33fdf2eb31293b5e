"""Host-side rotation-vector <-> matrix conversions for packing and write-back.

These replace the two ``cv2.Rodrigues`` call sites outside the residual loop
(``src/bundle_adjuster.py:157`` matrix->vector when packing, ``:235`` vector->matrix
when writing back).  They run once per camera per ``run()`` on the host; everything
per-observation happens on the GPU.  Semantics follow OpenCV's documented behaviour:
matrix->vector first projects onto the nearest rotation (SVD), vector->matrix returns
the identity below DBL_EPSILON.
"""
from __future__ import annotations

import numpy as np

_EPS = float(np.finfo(np.float64).eps)


def matrices_to_rvecs(Rs):
    """(N,3,3) -> (N,3).  A handful of matrices (a window's keyframes) go through the native walk extension (the same
    projection by Newton's polar iteration instead of an SVD, the same branches); many, or singular ones, through numpy."""
    Rs = np.ascontiguousarray(np.asarray(Rs, dtype=np.float64).reshape(-1, 3, 3))
    if 0 < Rs.shape[0] <= 64:
        try:
            from . import _mapwalk
            out = np.empty((Rs.shape[0], 3))
            if _mapwalk.rvecs_from_matrices(Rs, out) == Rs.shape[0]:
                return out
        except (ImportError, AttributeError):
            pass
    U, _, Vt = np.linalg.svd(Rs)
    Q = U @ Vt
    a = np.stack([Q[:, 2, 1] - Q[:, 1, 2], Q[:, 0, 2] - Q[:, 2, 0], Q[:, 1, 0] - Q[:, 0, 1]], axis=1)
    s = np.sqrt((a * a).sum(axis=1) * 0.25)
    c = np.clip((Q[:, 0, 0] + Q[:, 1, 1] + Q[:, 2, 2] - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    out = np.zeros((Q.shape[0], 3))
    reg = s >= 1e-5
    out[reg] = a[reg] * (theta[reg] / (2.0 * s[reg]))[:, None]
    for i in np.nonzero(~reg)[0]:
        if c[i] > 0:
            continue                                   # theta ~ 0
        q = Q[i]                                       # theta ~ pi: axis from the diagonal
        rx = np.sqrt(max((q[0, 0] + 1.0) * 0.5, 0.0))
        ry = np.sqrt(max((q[1, 1] + 1.0) * 0.5, 0.0)) * (-1.0 if q[0, 1] < 0 else 1.0)
        rz = np.sqrt(max((q[2, 2] + 1.0) * 0.5, 0.0)) * (-1.0 if q[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((q[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        out[i] = v * (theta[i] / np.linalg.norm(v))
    return out


def rvecs_to_matrices(rvecs):
    """(N,3) -> (N,3,3)."""
    r = np.asarray(rvecs, dtype=np.float64).reshape(-1, 3)
    theta = np.linalg.norm(r, axis=1)
    small = theta < _EPS
    th = np.where(small, 1.0, theta)
    k = r / th[:, None]
    c, s = np.cos(th), np.sin(th)
    kx = np.zeros((r.shape[0], 3, 3))
    kx[:, 0, 1], kx[:, 0, 2] = -k[:, 2], k[:, 1]
    kx[:, 1, 0], kx[:, 1, 2] = k[:, 2], -k[:, 0]
    kx[:, 2, 0], kx[:, 2, 1] = -k[:, 1], k[:, 0]
    R = (c[:, None, None] * np.eye(3) + (1.0 - c)[:, None, None] * (k[:, :, None] * k[:, None, :])
         + s[:, None, None] * kx)
    R[small] = np.eye(3)
    return R
