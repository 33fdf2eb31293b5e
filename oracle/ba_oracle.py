"""CPU oracle for the bundle-adjustment solve step.  TEST INFRASTRUCTURE ONLY.

Nothing under ``bundle_adjustment_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the
checker / the timed CPU baseline -- never as the thing shipped.

What it restates (reference = egirgin/bundle_adjustment, paths relative to the reference
root; ``scipy/`` = scipy 1.15.3 site-packages):

* ``reference_cost_function``  -- ``src/bundle_adjuster.py:24-72`` (per-observation loop,
  parameter layout ``[rvec(Na,3) | tvec(Na,3) | points(Np,3)]``, residual =
  observed - projected, x then y).
* ``reference_sparsity``       -- ``src/bundle_adjuster.py:74-120`` (0/1 pattern).
* ``rodrigues_to_mat`` / ``rodrigues_to_vec`` / ``project_point`` -- the two OpenCV calls
  the reference makes (``cv2.Rodrigues`` at ``src/bundle_adjuster.py:59,157,235`` and
  ``cv2.projectPoints(..., distCoeffs=None)`` at ``:67``).  OpenCV is a third-party
  dependency of the reference, unpinned (no requirements file) and NOT installed in the
  build container, so these follow OpenCV's published formulas (calib3d docs:
  Rodrigues' formula with the theta < DBL_EPSILON -> I branch; matrix->vector via SVD
  orthogonalisation + skew part; pinhole projection without distortion, z == 0 guarded
  as 1).  **Parity unpinned at the cv2 boundary**: no file of the reference holds a cv2
  output to check against.
* ``huber_rho``                -- ``scipy/optimize/_lsq/least_squares.py:169-178``.
* ``reference_least_squares``  -- the reference's solver call,
  ``src/bundle_adjuster.py:170-174``.

What is pinned: ``tests/golden/*.npz`` were produced by importing the reference's own
``src/bundle_adjuster.py`` / ``src/map_structures.py`` (unmodified, with numpy stand-ins
for the two cv2 calls -- see ``tests/golden/make_golden.py``) and record its
``_cost_function`` outputs, ``_prepare_sparsity_matrix`` pattern, ``run()`` log lines and
map write-back.  ``tests/test_oracle_golden.py`` checks this module against every one of
them.  So layout, ordering, sign, control flow and the scipy solver behaviour are
pinned by the reference itself; the projection arithmetic is pinned only against the
documented OpenCV formulas.

The second half of the file is the oracle for the parts the north star adds and the
reference has no code for (analytic Jacobian blocks, block normal equations, Schur
complement, PCG, LM).  They are checked against finite differences of the residual
above and against scipy / dense numpy linear algebra in ``tests/``.
"""
from __future__ import annotations

import numpy as np

DBL_EPSILON = float(np.finfo(np.float64).eps)


# ---------------------------------------------------------------------------
# cv2.Rodrigues / cv2.projectPoints restatement (published OpenCV formulas)
# ---------------------------------------------------------------------------
def rodrigues_to_mat(rvec):
    """cv2.Rodrigues(vector) -> 3x3.  theta < DBL_EPSILON gives the identity."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < DBL_EPSILON:
        return np.eye(3)
    c, s = np.cos(theta), np.sin(theta)
    k = r / theta
    kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * kx


def rodrigues_to_vec(R):
    """cv2.Rodrigues(matrix) -> rotation vector (3,).

    The input is first replaced by the nearest rotation U @ Vt of its SVD (so a
    slightly non-orthogonal product of recoverPose outputs is orthogonalised), the axis
    comes from the skew part, theta = acos((tr-1)/2); sin(theta) -> 0 has the published
    special cases (theta ~ 0 -> 0; theta ~ pi -> axis from the diagonal).
    """
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    rx = R[2, 1] - R[1, 2]
    ry = R[0, 2] - R[2, 0]
    rz = R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5
    c = min(1.0, max(-1.0, c))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1.0) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1.0) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1.0) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.sqrt(rx * rx + ry * ry + rz * rz))
    return np.array([rx, ry, rz]) * (theta / (2.0 * s))


def project_point(X, rvec, tvec, K):
    """cv2.projectPoints(X, rvec, tvec, K, None) for one point -> (2,)."""
    R = rodrigues_to_mat(rvec)
    t = np.asarray(tvec, dtype=np.float64).reshape(3)
    Xc = R @ np.asarray(X, dtype=np.float64).reshape(3) + t
    z = 1.0 / Xc[2] if Xc[2] != 0 else 1.0
    return np.array([Xc[0] * z * K[0, 0] + K[0, 2], Xc[1] * z * K[1, 1] + K[1, 2]])


# ---------------------------------------------------------------------------
# Reference restatement (a4, a7, a8 of SURVEY.md section 8)
# ---------------------------------------------------------------------------
def reference_cost_function(params, fixed_kf_pose, fixed_kf_id, adjustable_kf_ids,
                            map_point_ids, observations, keypoints_2d, camera_matrix):
    """Per-observation loop of src/bundle_adjuster.py:24-72, same control flow."""
    na = len(adjustable_kf_ids)
    npnt = len(map_point_ids)
    poses_rvec = params[0:na * 3].reshape((na, 3))
    poses_tvec = params[na * 3:na * 6].reshape((na, 3))
    points_3d = params[na * 6:].reshape((npnt, 3))
    adj = {kf: i for i, kf in enumerate(adjustable_kf_ids)}
    mpi = {mp: i for i, mp in enumerate(map_point_ids)}
    errors = []
    fixed_R, fixed_t = fixed_kf_pose
    for obs_kf_id, obs_mp_id in observations:
        mp_idx = mpi.get(obs_mp_id)
        if mp_idx is None:
            continue
        if obs_kf_id == fixed_kf_id:
            rvec = rodrigues_to_vec(fixed_R)          # :59 (re-done per observation)
            tvec = fixed_t
        else:
            kf_idx = adj.get(obs_kf_id)
            if kf_idx is None:
                continue
            rvec = poses_rvec[kf_idx]
            tvec = poses_tvec[kf_idx]
        proj = project_point(points_3d[mp_idx], rvec, tvec, camera_matrix)   # :67
        obs = np.asarray(keypoints_2d[(obs_kf_id, obs_mp_id)], dtype=np.float64)
        errors.extend((obs - proj).ravel())            # :68-69
    return np.array(errors)


def reference_sparsity(num_adj_kfs, num_mps, adj_kf_ids, mp_ids, observations):
    """0/1 pattern of src/bundle_adjuster.py:74-120 as COO (rows, cols) arrays,
    row-major sorted.  Rows 2i, 2i+1 <-> observation i."""
    adj = {kf: i for i, kf in enumerate(adj_kf_ids)}
    mpi = {mp: i for i, mp in enumerate(mp_ids)}
    rows, cols = [], []
    for i, (kf, mp) in enumerate(observations):
        m = mpi.get(mp)
        if m is None:
            continue
        c = []
        if kf in adj:
            k = adj[kf]
            c += [3 * k, 3 * k + 1, 3 * k + 2]
            c += [3 * num_adj_kfs + 3 * k + j for j in range(3)]
        c += [6 * num_adj_kfs + 3 * m + j for j in range(3)]
        for rr in (2 * i, 2 * i + 1):
            rows += [rr] * len(c)
            cols += sorted(c)
    return np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)


def huber_rho(z):
    """scipy/optimize/_lsq/least_squares.py:169-178 with f_scale = 1.
    z = f**2 per scalar residual.  Returns rho, rho', rho''."""
    z = np.asarray(z, dtype=np.float64)
    mask = z <= 1
    rho0 = np.where(mask, z, 2.0 * np.sqrt(np.where(mask, 1.0, z)) - 1.0)
    rho1 = np.where(mask, 1.0, np.where(mask, 1.0, z) ** -0.5)
    rho2 = np.where(mask, 0.0, -0.5 * np.where(mask, 1.0, z) ** -1.5)
    return rho0, rho1, rho2


def reference_least_squares(fun, x0, jac_sparsity, args=(), **overrides):
    """The reference's solver call, src/bundle_adjuster.py:170-174."""
    from scipy.optimize import least_squares
    kw = dict(jac_sparsity=jac_sparsity, loss='huber', args=args, verbose=0,
              xtol=1e-5, ftol=1e-5, max_nfev=50)
    kw.update(overrides)
    return least_squares(fun, x0, **kw)


# ---------------------------------------------------------------------------
# Flat (SoA) problem form used by the HIP library:  cams (Nc,6) = [rvec | tvec],
# pts (Np,3), obs (cam_idx, pt_idx, uv), intrinsics K4 = (fx, fy, cx, cy).
# ---------------------------------------------------------------------------
def rodrigues_batch(rvecs):
    """(N,3) -> (N,3,3), same branch as rodrigues_to_mat."""
    r = np.asarray(rvecs, dtype=np.float64).reshape(-1, 3)
    theta = np.sqrt((r * r).sum(axis=1))
    small = theta < DBL_EPSILON
    th = np.where(small, 1.0, theta)
    k = r / th[:, None]
    c, s = np.cos(th), np.sin(th)
    K = np.zeros((r.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    R = (c[:, None, None] * np.eye(3)[None] + (1.0 - c)[:, None, None] * k[:, :, None] * k[:, None, :]
         + s[:, None, None] * K)
    R[small] = np.eye(3)
    return R


def residuals(cams, pts, cam_idx, pt_idx, uv, K4):
    """Vectorised reprojection residuals, (Nobs,2): observed - projected."""
    cams = np.asarray(cams, dtype=np.float64).reshape(-1, 6)
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    R = rodrigues_batch(cams[:, :3])
    Xc = np.einsum('nij,nj->ni', R[cam_idx], pts[pt_idx]) + cams[cam_idx, 3:]
    z = np.where(Xc[:, 2] != 0, 1.0 / np.where(Xc[:, 2] != 0, Xc[:, 2], 1.0), 1.0)
    proj = np.stack([Xc[:, 0] * z * K4[0] + K4[2], Xc[:, 1] * z * K4[1] + K4[3]], axis=1)
    return np.asarray(uv, dtype=np.float64).reshape(-1, 2) - proj


def so3_right_jacobian(rvecs):
    """J_r(r) = I - b [r]x + d [r]x^2,  b = (1-cos t)/t^2,  d = (t - sin t)/t^3,
    so that R(r + e) = R(r) Exp(J_r(r) e) + O(e^2).  Closed form for t >= 0.05, Taylor
    series to t^6 below that (truncation error < 1e-16 there)."""
    r = np.asarray(rvecs, dtype=np.float64).reshape(-1, 3)
    t2 = (r * r).sum(axis=1)
    t = np.sqrt(t2)
    small = t < 0.05
    ts = np.where(small, 1.0, t)
    b = np.where(small, 0.5 - t2 / 24.0 + t2 * t2 / 720.0 - t2 ** 3 / 40320.0,
                 (1.0 - np.cos(ts)) / (ts * ts))
    d = np.where(small, 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 ** 3 / 362880.0,
                 (ts - np.sin(ts)) / (ts ** 3))
    rx = np.zeros((r.shape[0], 3, 3))
    rx[:, 0, 1], rx[:, 0, 2] = -r[:, 2], r[:, 1]
    rx[:, 1, 0], rx[:, 1, 2] = r[:, 2], -r[:, 0]
    rx[:, 2, 0], rx[:, 2, 1] = -r[:, 1], r[:, 0]
    return np.eye(3)[None] - b[:, None, None] * rx + d[:, None, None] * (rx @ rx)


def jacobian_blocks(cams, pts, cam_idx, pt_idx, K4):
    """Analytic d(residual)/d(cam) (Nobs,2,6) [rvec | tvec] and d(residual)/d(point)
    (Nobs,2,3) for the ADDITIVE rotation-vector parameterisation scipy uses
    (x + step, scipy/optimize/_lsq/trf.py:497-498):
        d(R X)/dr = -R [X]x J_r(r);   residual = observed - pi(R X + t)."""
    cams = np.asarray(cams, dtype=np.float64).reshape(-1, 6)
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    R = rodrigues_batch(cams[:, :3])[cam_idx]
    M = so3_right_jacobian(cams[:, :3])[cam_idx]
    X = pts[pt_idx]
    Xc = np.einsum('nij,nj->ni', R, X) + cams[cam_idx, 3:]
    iz = 1.0 / Xc[:, 2]
    n = X.shape[0]
    dpi = np.zeros((n, 2, 3))
    dpi[:, 0, 0] = K4[0] * iz
    dpi[:, 0, 2] = -K4[0] * Xc[:, 0] * iz * iz
    dpi[:, 1, 1] = K4[1] * iz
    dpi[:, 1, 2] = -K4[1] * Xc[:, 1] * iz * iz
    P = dpi @ R                                  # (n,2,3)
    Xx = np.zeros((n, 3, 3))
    Xx[:, 0, 1], Xx[:, 0, 2] = -X[:, 2], X[:, 1]
    Xx[:, 1, 0], Xx[:, 1, 2] = X[:, 2], -X[:, 0]
    Xx[:, 2, 0], Xx[:, 2, 1] = -X[:, 1], X[:, 0]
    Jc = np.concatenate([P @ Xx @ M, -dpi], axis=2)
    Jp = -P
    return Jc, Jp


def robust_weights(res, loss):
    """IRLS weights per scalar residual: rho'(f^2).  loss in {'linear','huber'}."""
    if loss == 'linear':
        return np.ones_like(res)
    return huber_rho(res * res)[1]


def robust_cost(res, loss):
    """0.5 * sum rho(f^2), the quantity scipy minimises (least_squares.py:899-901)."""
    z = res * res
    return 0.5 * float((z if loss == 'linear' else huber_rho(z)[0]).sum())


def normal_equations(cams, pts, cam_idx, pt_idx, uv, K4, fixed_cam=-1, loss='linear'):
    """Block normal equations of the weighted Gauss-Newton model
        Hcc (Nc,6,6), Hpp (Np,3,3), bc (Nc,6) = Jc^T w r, bp (Np,3) = Jp^T w r,
    plus per-observation W = Jc^T w Jp (Nobs,6,3).  The fixed camera's Jc is zero."""
    nc = np.asarray(cams).reshape(-1, 6).shape[0]
    npt = np.asarray(pts).reshape(-1, 3).shape[0]
    res = residuals(cams, pts, cam_idx, pt_idx, uv, K4)
    w = robust_weights(res, loss)
    Jc, Jp = jacobian_blocks(cams, pts, cam_idx, pt_idx, K4)
    if fixed_cam >= 0:
        Jc = Jc * (cam_idx != fixed_cam)[:, None, None]
    Jcw = Jc * w[:, :, None]
    Jpw = Jp * w[:, :, None]
    Hcc = np.zeros((nc, 6, 6))
    Hpp = np.zeros((npt, 3, 3))
    bc = np.zeros((nc, 6))
    bp = np.zeros((npt, 3))
    np.add.at(Hcc, cam_idx, np.einsum('nki,nkj->nij', Jcw, Jc))
    np.add.at(Hpp, pt_idx, np.einsum('nki,nkj->nij', Jpw, Jp))
    np.add.at(bc, cam_idx, np.einsum('nki,nk->ni', Jcw, res))
    np.add.at(bp, pt_idx, np.einsum('nki,nk->ni', Jpw, res))
    W = np.einsum('nki,nkj->nij', Jcw, Jp)
    return dict(Hcc=Hcc, Hpp=Hpp, bc=bc, bp=bp, W=W, res=res, w=w, Jc=Jc, Jp=Jp)


def sym6_pack(H):
    """(N,6,6) symmetric -> (N,21) upper triangle, row-major (00,01,..,05,11,..,55)."""
    iu = np.triu_indices(6)
    return H[:, iu[0], iu[1]]


def sym3_pack(H):
    iu = np.triu_indices(3)
    return H[:, iu[0], iu[1]]


def damp_blocks(H, lam, floor=1e-12):
    """Marquardt damping: H + lam * diag(max(diag(H), floor))."""
    Hd = H.copy()
    n = H.shape[1]
    d = np.maximum(H[:, np.arange(n), np.arange(n)], floor)
    Hd[:, np.arange(n), np.arange(n)] += lam * d
    return Hd


def schur_dense(ne, cam_idx, pt_idx, lam, fixed_cam=-1):
    """Dense reduced camera system for small problems:
        S = (Hcc + lam Dc) - sum_p W_p (Hpp + lam Dp)^-1 W_p^T ,
        rhs = -(bc - W (Hpp + lam Dp)^-1 bp).
    The fixed camera's row/column is replaced by identity / zero rhs."""
    nc = ne['Hcc'].shape[0]
    nb = ne['Hcc'].shape[1]
    npt = ne['Hpp'].shape[0]
    Hccd = damp_blocks(ne['Hcc'], lam)
    Hppinv = np.linalg.inv(damp_blocks(ne['Hpp'], lam))
    S = np.zeros((nb * nc, nb * nc))
    for c in range(nc):
        S[nb * c:nb * c + nb, nb * c:nb * c + nb] = Hccd[c]
    # W as a dense (nb Nc, 3Np) matrix
    Wd = np.zeros((nb * nc, 3 * npt))
    for o in range(len(cam_idx)):
        c, p = cam_idx[o], pt_idx[o]
        Wd[nb * c:nb * c + nb, 3 * p:3 * p + 3] += ne['W'][o]
    Hinv = np.zeros((3 * npt, 3 * npt))
    for p in range(npt):
        Hinv[3 * p:3 * p + 3, 3 * p:3 * p + 3] = Hppinv[p]
    S -= Wd @ Hinv @ Wd.T
    rhs = -(ne['bc'].ravel() - Wd @ Hinv @ ne['bp'].ravel())
    if fixed_cam >= 0:
        sl = slice(nb * fixed_cam, nb * fixed_cam + nb)
        S[sl, :] = 0
        S[:, sl] = 0
        S[sl, sl] = np.eye(nb)
        rhs[sl] = 0
    return S, rhs, Wd, Hinv


class SchurOperator:
    """Matrix-free S*v exactly as the device computes it (two passes over the
    observation list: by point, then by camera)."""

    def __init__(self, ne, cam_idx, pt_idx, lam, fixed_cam=-1):
        self.nc = ne['Hcc'].shape[0]
        self.nb = ne['Hcc'].shape[1]              # camera block size: 6 (the reference's pinhole) or 9 (BAL)
        self.np_ = ne['Hpp'].shape[0]
        self.cam_idx, self.pt_idx = cam_idx, pt_idx
        self.W = ne['W']
        self.Hccd = damp_blocks(ne['Hcc'], lam)
        self.Hppinv = np.linalg.inv(damp_blocks(ne['Hpp'], lam))
        self.fixed = fixed_cam
        self.bc, self.bp = ne['bc'], ne['bp']

    def wt_times(self, v):                     # (Nc,6) -> (Np,3):  Hpp^-1 W^T v
        u = np.zeros((self.np_, 3))
        np.add.at(u, self.pt_idx, np.einsum('nij,ni->nj', self.W, v[self.cam_idx]))
        return np.einsum('pij,pj->pi', self.Hppinv, u)

    def w_times(self, y):                      # (Np,3) -> (Nc,6):  W y
        q = np.zeros((self.nc, self.nb))
        np.add.at(q, self.cam_idx, np.einsum('nij,nj->ni', self.W, y[self.pt_idx]))
        return q

    def apply(self, v):
        v = v.reshape(self.nc, self.nb)
        q = np.einsum('cij,cj->ci', self.Hccd, v) - self.w_times(self.wt_times(v))
        if self.fixed >= 0:
            q[self.fixed] = v[self.fixed]
        return q

    def rhs(self):
        y0 = np.einsum('pij,pj->pi', self.Hppinv, self.bp)
        g = -(self.bc - self.w_times(y0))
        if self.fixed >= 0:
            g[self.fixed] = 0
        return g

    def schur_diag_blocks(self):
        """Block diagonal of S (Schur-Jacobi preconditioner)."""
        D = self.Hccd.copy()
        t = np.einsum('nij,njk,nlk->nil', self.W, self.Hppinv[self.pt_idx], self.W)
        np.subtract.at(D, self.cam_idx, t)
        if self.fixed >= 0:
            D[self.fixed] = np.eye(self.nb)
        return D

    def back_substitute(self, dc):
        """dp = -(Hpp+lam D)^-1 (bp + W^T dc)."""
        u = np.zeros((self.np_, 3))
        np.add.at(u, self.pt_idx, np.einsum('nij,ni->nj', self.W, dc[self.cam_idx]))
        return -np.einsum('pij,pj->pi', self.Hppinv, self.bp + u)


AGG = 16          # cameras per coarse aggregate (= one camera-vector workgroup of the device, csrc/ba_coarse.hpp)


def coarse_matrix(op, fixed_cam=-1):
    """E = P^T S P of the two-level preconditioner (csrc/ba_coarse.hpp): P piecewise constant over aggregates of AGG
    consecutive cameras, one coarse unknown per parameter; the fixed camera is left out.  Built from the blocks like
    the device does: sum of Hccd over an aggregate minus sum_p U_{p,a}^T Hpp^-1 U_{p,b}, U_{p,a} = sum of the point's
    W_o^T over its observations in aggregate a.  Returns E (6 Na, 6 Na)."""
    nc = op.Hccd.shape[0]
    na = (nc + AGG - 1) // AGG
    agg_of_obs = op.cam_idx // AGG
    free = op.cam_idx != fixed_cam
    # W_o^T (3x6) per observation, summed per (point, aggregate)
    key = op.pt_idx.astype(np.int64) * na + agg_of_obs
    uniq, inv = np.unique(key[free], return_inverse=True)
    U = np.zeros((uniq.shape[0], 3, 6))
    np.add.at(U, inv, np.transpose(op.W[free], (0, 2, 1)))
    pt_of = (uniq // na).astype(np.int64)
    ag_of = (uniq % na).astype(np.int64)
    E = np.zeros((na, 6, na, 6))
    for c in range(nc):
        if c != fixed_cam:
            E[c // AGG, :, c // AGG, :] += op.Hccd[c]
    T = np.einsum('rkl,rlj->rkj', op.Hppinv[pt_of], U)                      # Hinv U per (point, aggregate)
    # pairs of entries of the same point
    order = np.argsort(pt_of, kind='stable')
    starts = np.flatnonzero(np.r_[True, pt_of[order][1:] != pt_of[order][:-1], True])
    for s0, s1 in zip(starts[:-1], starts[1:]):
        idx = order[s0:s1]
        for i in idx:
            for j in idx:
                E[ag_of[i], :, ag_of[j], :] -= U[i].T @ T[j]
    E = E.reshape(6 * na, 6 * na)
    d = np.diag(E).copy()
    bad = ~(d > 0)
    E[bad, bad] = 1.0                                                       # (an aggregate holding only the fixed camera)
    return E


def two_level_apply(Minv, Einv, fixed_cam=-1):
    """r (Nc,6) -> M_J^-1 r + P E^-1 P^T r."""
    nc = Minv.shape[0]
    na = (nc + AGG - 1) // AGG
    agg = np.arange(nc) // AGG

    def apply(r):
        rr = r.copy()
        if fixed_cam >= 0:
            rr[fixed_cam] = 0
        rc = np.zeros((na, 6))
        np.add.at(rc, agg, rr)
        zc = (Einv @ rc.ravel()).reshape(na, 6)
        z = np.einsum('cij,cj->ci', Minv, r) + zc[agg]
        if fixed_cam >= 0:
            z[fixed_cam] = np.einsum('ij,j->i', Minv[fixed_cam], r[fixed_cam])
        return z
    return apply


def pcg(op, rhs, Minv, tol, max_iters, min_iters=0, model_tol=0.0, model_min_iters=5):
    """Preconditioned CG on S x = rhs with block-Jacobi Minv (Nc,6,6), or any callable r -> M^-1 r.
    Stops when sqrt(rz / rz0) <= tol (after min_iters), at max_iters, or -- model_tol > 0 -- by Nash & Sofer's
    truncated-Newton test on the quadratic model q(x) = 1/2 x^T S x - rhs^T x that CG minimises: iteration i lowers q by
    1/2 alpha_i (r.z)_{i-1}; stop after iteration i >= model_min_iters when i times that is <= model_tol of the whole decrease
    (the device: ba_options.pcg_model_tol, k_pcg_step).  Returns x (Nc,6), iterations, final residual vector."""
    if not callable(Minv):
        blocks = Minv
        Minv = lambda r_: np.einsum('cij,cj->ci', blocks, r_)          # noqa: E731
    x = np.zeros_like(rhs)
    r = rhs.copy()
    z = Minv(r)
    p = z.copy()
    rz = float((r * z).sum())
    rz0 = rz
    it = 0
    q_tot = 0.0
    if rz0 <= 0:
        return x, 0, r
    while it < max_iters:
        q = op.apply(p)
        pq = float((p * q).sum())
        if not (pq > 0):
            break
        alpha = rz / pq
        x += alpha * p
        r -= alpha * q
        z = Minv(r)
        rz_new = float((r * z).sum())
        it += 1
        dq = 0.5 * alpha * rz
        q_tot += dq
        if it >= min_iters and rz_new <= tol * tol * rz0:
            rz = rz_new
            break
        if model_tol > 0 and it >= model_min_iters and it * dq <= model_tol * q_tot:
            rz = rz_new
            break
        beta = rz_new / rz
        rz = rz_new
        p = z + beta * p
    return x, it, r


def band_structured(cam_idx, pt_idx, n_cams):
    """ba_set_problem's band statistic: the mean camera span (largest minus smallest camera index) of a track is at most
    n_cams / 8 -- sequential captures, where a landmark is seen from a few neighbouring cameras."""
    cam_idx, pt_idx = np.asarray(cam_idx), np.asarray(pt_idx)
    if cam_idx.size == 0:
        return False
    npts = int(pt_idx.max()) + 1
    lo = np.full(npts, n_cams, dtype=np.int64)
    hi = np.full(npts, -1, dtype=np.int64)
    np.minimum.at(lo, pt_idx, cam_idx)
    np.maximum.at(hi, pt_idx, cam_idx)
    seen = hi >= 0
    return bool(seen.any() and float((hi[seen] - lo[seen]).sum()) / int(seen.sum()) <= n_cams / 8.0)


def lm_solve(cams, pts, cam_idx, pt_idx, uv, K4, fixed_cam=-1, loss='linear',
             max_iters=50, ftol=1e-10, xtol=1e-10, gtol=1e-10, lam0=1e-4,
             pcg_tol=1e-1, pcg_max_iters=200, precond='schur_jacobi', verbose=False, linear_solver='pcg', model='pinhole',
             pcg_model_tol=-1.0, pcg_model_min_iters=5, precond_lag=3, cap_floor=True):
    """CPU mirror of the device LM / Schur / PCG loop (same formulas, same update
    rules, same stopping tests) -- see ba_solve in bundle_adjustment_amd/csrc/ba_hip.hip.
    precond_lag (Schur-Jacobi only; the device's ba_options.precond_lag, same default and same rule): up to that many
    consecutive damped systems keep the preconditioner blocks built for an earlier one, unless the damping has moved by more
    than 10x since the build, the last inner solve needed more than 1.5x + 2 the iterations of the first one after it, or the
    last accepted step lowered the cost by more than 1 % (a re-damped system after a rejected step may always keep them).
    linear_solver='dense' solves the explicit reduced system (schur_dense) exactly instead, as the
    single-launch solver for window-sized problems does (csrc/ba_small.hpp).
    model='bal': the 9-parameter BAL camera [rvec | t | f k1 k2] (bal_residuals, bal_normal_equations; K4 is ignored) --
    the mirror of ba_solve_bal (the BalCam instantiation of the loop, csrc/ba_models.hpp); same loop, 9x9 camera blocks.
    Returns dict(cams, pts, iterations, accepted, sse0, sse, cost0, cost, pcg_iters,
    history)."""
    nb = 9 if model == 'bal' else 6
    if pcg_model_tol < 0:           # the device's automatic default (ba_options.pcg_model_tol = -1): on for band-structured
        # problems solved to a loose outer tolerance (ftol >= 1e-6)
        pcg_model_tol = 0.5 if (ftol >= 1e-6 and band_structured(cam_idx, pt_idx, np.asarray(cams).reshape(-1, nb).shape[0])) else 0.0
    cams = np.array(cams, dtype=np.float64).reshape(-1, nb)
    pts = np.array(pts, dtype=np.float64).reshape(-1, 3)
    lam, nu = lam0, 2.0
    if model == 'bal':
        def residuals(c_, p_, ci_, pi_, uv_, K4_):                       # noqa: F811  (shadows the pinhole residual)
            return bal_residuals(c_, p_, ci_, pi_, uv_)

        def normal_equations(c_, p_, ci_, pi_, uv_, K4_, fixed_, loss_):  # noqa: F811
            return bal_normal_equations(c_, p_, ci_, pi_, uv_, fixed_, loss_)
    else:
        residuals, normal_equations = globals()['residuals'], globals()['normal_equations']
    res = residuals(cams, pts, cam_idx, pt_idx, uv, K4)
    cost = robust_cost(res, loss)
    sse0 = float((res * res).sum())
    cost0 = cost
    hist = []
    it = acc = pcg_total = 0
    status = 'max_iters'
    lag = precond_lag if (precond == 'schur_jacobi' and linear_solver != 'dense') else 0
    Minv_kept, lam_built, kept, pcg_at_build, pcg_last = None, 0.0, 0, -1, -1
    last_decrease, fresh = 1.0, True
    lam_floor = 0.0                     # cap-aware damping (ba_solve): 3 x the damping at which an inner solve last hit pcg_max_iters
    while it < max_iters:
        ne = normal_equations(cams, pts, cam_idx, pt_idx, uv, K4, fixed_cam, loss)
        gmax = max(np.abs(ne['bc']).max(), np.abs(ne['bp']).max())
        if gmax <= gtol:
            status = 'gtol'
            break
        op = SchurOperator(ne, cam_idx, pt_idx, lam, fixed_cam)
        rhs = op.rhs()
        keep = (lag > 0 and Minv_kept is not None and kept < lag and lam <= 10.0 * lam_built and lam >= 0.1 * lam_built
                and (pcg_at_build < 0 or pcg_last <= pcg_at_build + pcg_at_build // 2 + 2)
                and (not fresh or last_decrease <= 1e-2))
        if keep:
            Minv = Minv_kept
            kept += 1
        else:
            D = op.Hccd.copy() if precond == 'jacobi' else op.schur_diag_blocks()
            if fixed_cam >= 0:
                D[fixed_cam] = np.eye(nb)
            Minv = np.linalg.inv(D)
            Minv_kept, lam_built, kept, pcg_at_build = (Minv if precond == 'schur_jacobi' else None), lam, 0, -1
        if precond == 'two_level':
            Minv = two_level_apply(Minv, np.linalg.inv(coarse_matrix(op, fixed_cam)), fixed_cam)
        if linear_solver == 'dense':
            S_, g_, _, _ = schur_dense(ne, cam_idx, pt_idx, lam, fixed_cam)
            dc, k, rfin = np.linalg.solve(S_, g_).reshape(-1, nb), 0, np.zeros_like(rhs)
        else:
            dc, k, rfin = pcg(op, rhs, Minv, pcg_tol, pcg_max_iters, model_tol=pcg_model_tol, model_min_iters=pcg_model_min_iters)
        pcg_total += k
        if cap_floor and linear_solver != 'dense' and k >= pcg_max_iters:
            lam_floor = max(lam_floor, 3.0 * lam)
        pcg_last = k
        if pcg_at_build < 0:
            pcg_at_build = k
        dp = op.back_substitute(dc)
        # model decrease of the damped, inexactly solved system (LM section)
        dcd = np.maximum(ne['Hcc'][:, np.arange(nb), np.arange(nb)], 1e-12)
        dpd = np.maximum(ne['Hpp'][:, np.arange(3), np.arange(3)], 1e-12)
        if fixed_cam >= 0:
            dcd[fixed_cam] = 0
        gTd = float((ne['bc'] * dc).sum() + (ne['bp'] * dp).sum())
        dDd = float((dcd * dc * dc).sum() + (dpd * dp * dp).sum())
        model = 0.5 * (lam * dDd - gTd + float((dc * rfin).sum()))
        cams_new, pts_new = cams + dc, pts + dp
        res_new = residuals(cams_new, pts_new, cam_idx, pt_idx, uv, K4)
        cost_new = robust_cost(res_new, loss)
        rho = (cost - cost_new) / model if model > 0 else -1.0
        step = np.sqrt(float((dc * dc).sum() + (dp * dp).sum()))
        xnorm = np.sqrt(float((cams * cams).sum() + (pts * pts).sum()))
        it += 1
        hist.append(dict(it=it, cost=cost, cost_new=cost_new, lam=lam, rho=rho, pcg=k, step=step))
        if verbose:
            print(hist[-1])
        if rho > 0 and np.isfinite(cost_new):
            dcost = cost - cost_new
            last_decrease = dcost / cost_new if cost_new > 0 else 1.0
            fresh = True
            cams, pts, cost = cams_new, pts_new, cost_new
            acc += 1
            lam = lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)
            lam = max(lam, 1e-12, lam_floor)
            nu = 2.0
            if dcost <= ftol * cost:
                status = 'ftol'
                break
        else:
            lam = min(lam * nu, 1e12)
            nu *= 2.0
            fresh = False                  # the same linearisation, damped again
        if step <= xtol * (xtol + xnorm):
            status = 'xtol'
            break
    res = residuals(cams, pts, cam_idx, pt_idx, uv, K4)
    return dict(cams=cams, pts=pts, iterations=it, accepted=acc, sse0=sse0,
                sse=float((res * res).sum()), cost0=cost0, cost=cost, pcg_iters=pcg_total,
                status=status, history=hist, lam=lam)


# ---------------------------------------------------------------------------
# scipy-side helpers used by parity / convergence tests and the CPU baseline
# ---------------------------------------------------------------------------
def pack_reference_params(cams, pts, fixed_cam):
    """[rvec(Na,3) | tvec(Na,3) | points(Np,3)] of src/bundle_adjuster.py:157-162,
    adjustable cameras in index order with `fixed_cam` left out."""
    cams = np.asarray(cams).reshape(-1, 6)
    adj = [i for i in range(cams.shape[0]) if i != fixed_cam]
    return np.concatenate([cams[adj, :3].ravel(), cams[adj, 3:].ravel(), np.asarray(pts).ravel()]), adj


def unpack_reference_params(x, cams0, fixed_cam, npts):
    cams = np.array(cams0, dtype=np.float64).reshape(-1, 6)
    adj = [i for i in range(cams.shape[0]) if i != fixed_cam]
    na = len(adj)
    cams[adj, :3] = x[:3 * na].reshape(na, 3)
    cams[adj, 3:] = x[3 * na:6 * na].reshape(na, 3)
    return cams, x[6 * na:].reshape(npts, 3).copy()


def flat_sparsity(nc, npts, cam_idx, pt_idx, fixed_cam):
    """scipy CSR 0/1 pattern equal to reference_sparsity for the flat problem."""
    from scipy.sparse import coo_matrix
    adj = [i for i in range(nc) if i != fixed_cam]
    amap = -np.ones(nc, dtype=np.int64)
    amap[adj] = np.arange(len(adj))
    na = len(adj)
    nobs = len(cam_idx)
    rows, cols = [], []
    k = amap[cam_idx]
    isadj = k >= 0
    for rr in (0, 1):
        r_all = 2 * np.arange(nobs) + rr
        for j in range(3):
            rows.append(r_all[isadj]); cols.append(3 * k[isadj] + j)
            rows.append(r_all[isadj]); cols.append(3 * na + 3 * k[isadj] + j)
            rows.append(r_all); cols.append(6 * na + 3 * pt_idx + j)
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    return coo_matrix((np.ones(len(rows), dtype=int), (rows, cols)),
                      shape=(2 * nobs, 6 * na + 3 * npts)).tocsr()


def flat_residual_fun(cams0, npts, cam_idx, pt_idx, uv, K4, fixed_cam):
    """fun(x) on the reference parameter layout, vectorised (the 'strong' CPU path)."""
    def fun(x):
        cams, pts = unpack_reference_params(x, cams0, fixed_cam, npts)
        return residuals(cams, pts, cam_idx, pt_idx, uv, K4).ravel()
    return fun


def flat_jacobian_fun(cams0, npts, cam_idx, pt_idx, K4, fixed_cam):
    """Analytic sparse Jacobian on the reference parameter layout (CSR)."""
    from scipy.sparse import coo_matrix
    nc = np.asarray(cams0).reshape(-1, 6).shape[0]
    adj = [i for i in range(nc) if i != fixed_cam]
    amap = -np.ones(nc, dtype=np.int64)
    amap[adj] = np.arange(len(adj))
    na = len(adj)
    nobs = len(cam_idx)
    k = amap[cam_idx]
    isadj = k >= 0

    def jac(x):
        cams, pts = unpack_reference_params(x, cams0, fixed_cam, npts)
        Jc, Jp = jacobian_blocks(cams, pts, cam_idx, pt_idx, K4)
        rows, cols, vals = [], [], []
        for rr in (0, 1):
            r_all = 2 * np.arange(nobs) + rr
            for j in range(3):
                rows.append(r_all[isadj]); cols.append(3 * k[isadj] + j); vals.append(Jc[isadj, rr, j])
                rows.append(r_all[isadj]); cols.append(3 * na + 3 * k[isadj] + j); vals.append(Jc[isadj, rr, 3 + j])
                rows.append(r_all); cols.append(6 * na + 3 * pt_idx + j); vals.append(Jp[:, rr, j])
        return coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                          shape=(2 * nobs, 6 * na + 3 * npts)).tocsr()
    return jac


# ---------------------------------------------------------------------------
# Two-view triangulation (row f3): restatement of src/pipeline.py:315-336
# ---------------------------------------------------------------------------
def triangulate_points(K, R_rel, t_rel, pts1, pts2):
    """``_triangulate_points`` of the reference with ``cv2.triangulatePoints`` restated from OpenCV's published DLT
    (per point the 4x4 system A = [x1 P1[2]-P1[0]; y1 P1[2]-P1[1]; x2 P2[2]-P2[0]; y2 P2[2]-P2[1]], solution = right
    singular vector of the smallest singular value; numpy SVD on A itself, not on A^T A).  The singular vector's sign
    is fixed as w >= 0 (OpenCV leaves it to its SVD; the reference's ``+ 1e-6`` makes the result depend on it at the
    1e-6 level: parity unpinned at the cv2 boundary).  Returns (xyz (n,3) for EVERY point, valid mask (n,))."""
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    R = np.asarray(R_rel, dtype=np.float64).reshape(3, 3)
    t = np.asarray(t_rel, dtype=np.float64).reshape(3, 1)
    P1 = K @ np.hstack((np.eye(3), np.zeros((3, 1))))
    P2 = K @ np.hstack((R, t))
    p1 = np.asarray(pts1, dtype=np.float64).reshape(-1, 2)
    p2 = np.asarray(pts2, dtype=np.float64).reshape(-1, 2)
    A = np.stack([p1[:, :1] * P1[2] - P1[0], p1[:, 1:] * P1[2] - P1[1],
                  p2[:, :1] * P2[2] - P2[0], p2[:, 1:] * P2[2] - P2[1]], axis=1)          # (n,4,4)
    _, _, Vt = np.linalg.svd(A)
    X = Vt[:, 3, :]
    X = np.where(X[:, 3:] < 0, -X, X)
    xyz = X[:, :3] / (X[:, 3:] + 1e-6)
    z2 = (xyz @ R.T + t.ravel())[:, 2]
    return xyz, (xyz[:, 2] > 0) & (z2 > 0)


# ---------------------------------------------------------------------------
# BAL 9-parameter camera (row f2): [rvec | t | f k1 k2], p = -P[:2]/P[2], proj = f (1 + k1 |p|^2 + k2 |p|^4) p
# ---------------------------------------------------------------------------
def bal_residuals(cams9, pts, cam_idx, pt_idx, uv):
    """observed - projected for the BAL camera model (grail.cs.washington.edu/projects/bal), (Nobs, 2)."""
    cams9 = np.asarray(cams9, dtype=np.float64).reshape(-1, 9)
    R = rodrigues_batch(cams9[:, :3])
    P = np.einsum('nij,nj->ni', R[cam_idx], np.asarray(pts, dtype=np.float64)[pt_idx]) + cams9[cam_idx, 3:6]
    p = -P[:, :2] / P[:, 2:3]
    n2 = (p * p).sum(axis=1)
    f, k1, k2 = cams9[cam_idx, 6], cams9[cam_idx, 7], cams9[cam_idx, 8]
    rad = 1.0 + k1 * n2 + k2 * n2 * n2
    return np.asarray(uv, dtype=np.float64) - (f * rad)[:, None] * p


def bal_jacobian_blocks(cams9, pts, cam_idx, pt_idx):
    """Analytic blocks of the BAL residual: Jc (Nobs, 2, 9) w.r.t. [rvec | t | f k1 k2] (additive rotation-vector update,
    like the 6-parameter blocks above) and Jp (Nobs, 2, 3) w.r.t. the point."""
    cams9 = np.asarray(cams9, dtype=np.float64).reshape(-1, 9)
    pts = np.asarray(pts, dtype=np.float64)
    R = rodrigues_batch(cams9[:, :3])[cam_idx]
    M = so3_right_jacobian(cams9[:, :3])[cam_idx]
    X = pts[pt_idx]
    P = np.einsum('nij,nj->ni', R, X) + cams9[cam_idx, 3:6]
    iz = 1.0 / P[:, 2]
    p = -P[:, :2] * iz[:, None]
    n2 = (p * p).sum(axis=1)
    f, k1, k2 = cams9[cam_idx, 6], cams9[cam_idx, 7], cams9[cam_idx, 8]
    rad = 1.0 + k1 * n2 + k2 * n2 * n2
    drad = k1 + 2.0 * k2 * n2                                   # d rad / d n2
    n = len(cam_idx)
    # d proj / d p  (2x2) = f (rad I + 2 drad p p^T)
    dproj_dp = f[:, None, None] * (rad[:, None, None] * np.eye(2)[None] + 2.0 * drad[:, None, None] * p[:, :, None] * p[:, None, :])
    # d p / d P (2x3) = -1/Pz [I | -P[:2]/Pz] = [-iz 0 Px iz^2; 0 -iz Py iz^2]
    dp_dP = np.zeros((n, 2, 3))
    dp_dP[:, 0, 0] = -iz
    dp_dP[:, 1, 1] = -iz
    dp_dP[:, 0, 2] = P[:, 0] * iz * iz
    dp_dP[:, 1, 2] = P[:, 1] * iz * iz
    dproj_dP = np.einsum('nij,njk->nik', dproj_dp, dp_dP)      # (n,2,3)
    Jp = -np.einsum('nij,njk->nik', dproj_dP, R)               # residual = uv - proj
    Xx = np.zeros((n, 3, 3))
    Xx[:, 0, 1], Xx[:, 0, 2] = -X[:, 2], X[:, 1]
    Xx[:, 1, 0], Xx[:, 1, 2] = X[:, 2], -X[:, 0]
    Xx[:, 2, 0], Xx[:, 2, 1] = -X[:, 1], X[:, 0]
    # d (R X) / d rvec = -R [X]x M
    dP_dr = -np.einsum('nij,njk,nkl->nil', R, Xx, M)
    Jc = np.zeros((n, 2, 9))
    Jc[:, :, :3] = -np.einsum('nij,njk->nik', dproj_dP, dP_dr)
    Jc[:, :, 3:6] = -dproj_dP
    Jc[:, :, 6] = -(rad[:, None] * p)
    Jc[:, :, 7] = -(f * n2)[:, None] * p
    Jc[:, :, 8] = -(f * n2 * n2)[:, None] * p
    return Jc, Jp


def bal_normal_equations(cams9, pts, cam_idx, pt_idx, uv, fixed_cam=-1, loss='linear'):
    """normal_equations for the BAL camera: Hcc (Nc,9,9), Hpp (Np,3,3), bc (Nc,9), bp (Np,3), W (Nobs,9,3)."""
    nc = np.asarray(cams9).reshape(-1, 9).shape[0]
    npt = np.asarray(pts).reshape(-1, 3).shape[0]
    res = bal_residuals(cams9, pts, cam_idx, pt_idx, uv)
    w = robust_weights(res, loss)
    Jc, Jp = bal_jacobian_blocks(cams9, pts, cam_idx, pt_idx)
    if fixed_cam >= 0:
        Jc = Jc * (np.asarray(cam_idx) != fixed_cam)[:, None, None]
    Jcw = Jc * w[:, :, None]
    Jpw = Jp * w[:, :, None]
    Hcc = np.zeros((nc, 9, 9))
    Hpp = np.zeros((npt, 3, 3))
    bc = np.zeros((nc, 9))
    bp = np.zeros((npt, 3))
    np.add.at(Hcc, cam_idx, np.einsum('nki,nkj->nij', Jcw, Jc))
    np.add.at(Hpp, pt_idx, np.einsum('nki,nkj->nij', Jpw, Jp))
    np.add.at(bc, cam_idx, np.einsum('nki,nk->ni', Jcw, res))
    np.add.at(bp, pt_idx, np.einsum('nki,nk->ni', Jpw, res))
    W = np.einsum('nki,nkj->nij', Jcw, Jp)
    return dict(Hcc=Hcc, Hpp=Hpp, bc=bc, bp=bp, W=W, res=res, w=w, Jc=Jc, Jp=Jp)
