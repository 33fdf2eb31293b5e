"""BAL ("Bundle Adjustment in the Large") problems: text reader / writer and the 9-parameter BAL camera
(SURVEY.md section 8f row 2).  The reference itself only has the shared-intrinsics pinhole of
``cv2.projectPoints(..., distCoeffs=None)`` (``src/bundle_adjuster.py:67``); BAL is the public format BASELINE config 5
("BAL-style Ladybug 1723-cam / 156k-point problem") is stated in.

File format (Agarwal et al., grail.cs.washington.edu/projects/bal)::

    <num_cameras> <num_points> <num_observations>
    <camera_index> <point_index> <x> <y>            one line per observation
    <camera parameter>                              9 lines per camera: rvec(3) t(3) f k1 k2
    <point coordinate>                              3 lines per point

Camera model: ``P = R(rvec) X + t;  p = -P[:2] / P[2]`` (the camera looks down -z);
``r = 1 + k1 |p|^2 + k2 |p|^4``; projection ``f r p`` (origin at the image centre).  Residual = observed - projected, x
then y, the reference's sign (``src/bundle_adjuster.py:68-69``).

What runs where: ``read_bal`` / ``write_bal`` are host code; ``Solver.residuals_bal`` evaluates the BAL residual on the
GPU (``ba_residuals_bal``: K1 with per-camera ``f, k1, k2``); ``to_pinhole`` converts a BAL problem whose cameras share
one focal length and have no distortion into the reference's model (z flipped, shared K), which the LM / Schur / PCG
solver then adjusts as it stands; ``solve`` / ``Solver.solve_bal`` adjust the full 9-parameter cameras (``ba_solve_bal``:
the same kernels and host loop as ``ba_solve``, instantiated for the ``BalCam`` model of ``csrc/ba_models.hpp`` -- LM + Schur
+ PCG with 2x9 camera blocks, ``jacobian_precision`` and multi-rank jobs included; checked step by step against
``oracle.lm_solve(model='bal')``, which is test infrastructure and never imported here), ``Solver.linearize_bal`` returns
the block normal equations (``ba_linearize_bal``).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .problem import BAProblem


@dataclass
class BALProblem:
    cams: np.ndarray        # (Nc,9) float64  [rvec | tvec | f k1 k2]
    pts: np.ndarray         # (Np,3) float64
    cam_idx: np.ndarray     # (Nobs,) int32
    pt_idx: np.ndarray      # (Nobs,) int32
    uv: np.ndarray          # (Nobs,2) float64, origin at the image centre

    @property
    def n_cams(self):
        return int(self.cams.shape[0])

    @property
    def n_pts(self):
        return int(self.pts.shape[0])

    @property
    def n_obs(self):
        return int(self.cam_idx.shape[0])

    def validate(self):
        if self.cams.ndim != 2 or self.cams.shape[1] != 9 or self.pts.ndim != 2 or self.pts.shape[1] != 3:
            raise ValueError("cams must be (Nc,9) and pts (Np,3)")
        if self.pt_idx.shape != (self.n_obs,) or self.uv.shape != (self.n_obs, 2):
            raise ValueError("observation arrays disagree in length")
        if self.n_obs and (self.cam_idx.min() < 0 or self.cam_idx.max() >= self.n_cams or
                           self.pt_idx.min() < 0 or self.pt_idx.max() >= self.n_pts):
            raise ValueError("observation index out of range")
        return self


def read_bal(path) -> BALProblem:
    """Parse a BAL text file (plain or .bz2 / .gz).  Whitespace-separated numbers; line structure is not required."""
    import bz2
    import gzip
    opener = bz2.open if str(path).endswith(".bz2") else gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        head = f.readline().split()
        while len(head) < 3:                                   # header split over lines
            more = f.readline()
            if not more:
                raise ValueError("truncated BAL header")
            head += more.split()
        nc, npt, nobs = int(head[0]), int(head[1]), int(head[2])
        rest = np.array(head[3:] + f.read().split(), dtype=np.float64)
    need = 4 * nobs + 9 * nc + 3 * npt
    if rest.size != need:
        raise ValueError(f"BAL file holds {rest.size} numbers after the header, {need} expected for "
                         f"{nc} cameras / {npt} points / {nobs} observations")
    obs = rest[:4 * nobs].reshape(nobs, 4)
    if nobs and (np.any(obs[:, :2] != np.floor(obs[:, :2]))):
        raise ValueError("non-integer camera / point index in the observation block")
    cams = rest[4 * nobs:4 * nobs + 9 * nc].reshape(nc, 9).copy()
    pts = rest[4 * nobs + 9 * nc:].reshape(npt, 3).copy()
    return BALProblem(cams, pts, obs[:, 0].astype(np.int32), obs[:, 1].astype(np.int32), obs[:, 2:].copy()).validate()


def write_bal(path, prob: BALProblem):
    """Write the BAL text format; numbers with 17 significant digits (``read_bal(write_bal(p))`` is exact)."""
    prob.validate()
    with open(path, "w") as f:
        f.write(f"{prob.n_cams} {prob.n_pts} {prob.n_obs}\n")
        for c, p, (x, y) in zip(prob.cam_idx.tolist(), prob.pt_idx.tolist(), prob.uv.tolist()):
            f.write(f"{c} {p}     {x!r} {y!r}\n")
        for v in prob.cams.ravel().tolist():
            f.write(f"{v!r}\n")
        for v in prob.pts.ravel().tolist():
            f.write(f"{v!r}\n")


def to_pinhole(prob: BALProblem, tol=0.0) -> BAProblem:
    """The same problem in the reference's camera model, possible when every BAL camera has the same focal length
    and no distortion (|k1|, |k2| <= tol): BAL projects ``-f P / P.z``; with ``S = diag(1, -1, -1)`` the camera
    ``(S R, S t)`` looking down +z with ``K4 = (f, -f, 0, 0)`` (a negative fy stands for BAL's y axis) gives
    ``u = f X'/Z' ... `` identical pixels.  Raises when the cameras do not allow it."""
    from .rotations import matrices_to_rvecs, rvecs_to_matrices
    f = prob.cams[:, 6]
    if np.ptp(f) > tol * max(1.0, abs(f[0])) or np.abs(prob.cams[:, 7:]).max() > tol:
        raise ValueError("to_pinhole needs one shared focal length and zero distortion")
    S = np.diag([1.0, -1.0, -1.0])
    R = S @ rvecs_to_matrices(prob.cams[:, :3])
    t = prob.cams[:, 3:6] @ S.T
    cams = np.concatenate([matrices_to_rvecs(R), t], axis=1)
    # BAL: u = -f Px/Pz, v = -f Py/Pz.  With P' = S P: Px' = Px, Py' = -Py, Pz' = -Pz  ->  u = f Px'/Pz', v = -f Py'/Pz'
    K4 = np.array([f[0], -f[0], 0.0, 0.0])
    return BAProblem(cams, prob.pts.copy(), prob.cam_idx.copy(), prob.pt_idx.copy(), prob.uv.copy(), K4, 0).validate()


def from_pinhole(prob: BAProblem) -> BALProblem:
    """A problem in the reference's camera model written as a BAL problem: needs |fy| = fx (BAL has one focal length);
    the principal point is taken out of the pixels (BAL's origin is the image centre) and, for fy = +fx, the y axis is
    turned over (BAL's camera looks down -z: v_bal = -(v - cy)).  Inverse of ``to_pinhole`` when K4 = (f, -f, 0, 0)."""
    from .rotations import matrices_to_rvecs, rvecs_to_matrices
    fx, fy, cx, cy = (float(v) for v in prob.K4)
    if abs(fy) != fx:
        raise ValueError("only |fy| = fx maps onto the BAL camera (one focal length)")
    S = np.diag([1.0, -1.0, -1.0])
    R = S @ rvecs_to_matrices(prob.cams[:, :3])
    t = prob.cams[:, 3:6] @ S.T
    cams = np.concatenate([matrices_to_rvecs(R), t, np.tile([fx, 0.0, 0.0], (prob.n_cams, 1))], axis=1)
    uv = prob.uv - np.array([cx, cy])
    if fy > 0:
        uv = uv * np.array([1.0, -1.0])
    return BALProblem(cams, prob.pts.copy(), prob.cam_idx.copy(), prob.pt_idx.copy(), uv).validate()


def solve(prob: BALProblem, device=0, fixed_cam=-1, **options):
    """Adjust a BAL problem on the GPU (``ba_solve_bal``: poses, points AND f / k1 / k2 per camera).  Returns
    ``(BALProblem with the adjusted parameters, summary dict)``; options as ``hip_backend.Solver.solve``."""
    from . import hip_backend
    with hip_backend.Solver(device) as s:
        summary, cams, pts = s.solve_bal(prob, fixed_cam=fixed_cam, **options)
    return BALProblem(cams, pts, prob.cam_idx.copy(), prob.pt_idx.copy(), prob.uv.copy()), summary
