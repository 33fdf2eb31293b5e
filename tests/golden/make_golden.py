#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's own code.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py

The reference's ``src/bundle_adjuster.py`` and ``src/map_structures.py`` are loaded
unmodified from /root/reference/src.  Three of their imports are not installed here, so
scratch modules are written to a temporary directory placed first on ``sys.path``:

* ``cv2``      -- only ``Rodrigues`` and ``projectPoints`` (the two symbols
                  bundle_adjuster.py touches, lines 59/67/157/235), scalar numpy code
                  written from OpenCV's documented formulas, independent of oracle/.
* ``open3d``   -- empty point-cloud stub (``has_points()`` False, so no PCD is written).
* ``visualization`` -- ``plot_and_save_sparsity`` no-op.

Consequence, stated wherever these vectors are used: they pin the reference's
parameter layout, observation/row order, residual sign, window selection, skip /
divergence / write-back behaviour, log lines and (through the real scipy) the solver
trajectory -- but NOT OpenCV's own arithmetic (parity unpinned at the cv2 boundary).

Nothing from the reference is copied into the repo: the .npz files hold inputs and
outputs only.
"""
import contextlib
import io
import os
import sys
import tempfile
import textwrap

import numpy as np

REF_SRC = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))

CV2_STANDIN = '''
import math
import numpy as np
DBL_EPSILON = 2.220446049250313e-16

def _vec2mat(r):
    rx, ry, rz = float(r[0]), float(r[1]), float(r[2])
    th = math.sqrt(rx*rx + ry*ry + rz*rz)
    if th < DBL_EPSILON:
        return np.eye(3)
    c, s = math.cos(th), math.sin(th)
    c1 = 1.0 - c
    x, y, z = rx/th, ry/th, rz/th
    return np.array([[c + c1*x*x,   c1*x*y - s*z, c1*x*z + s*y],
                     [c1*x*y + s*z, c + c1*y*y,   c1*y*z - s*x],
                     [c1*x*z - s*y, c1*y*z + s*x, c + c1*z*z]])

def _mat2vec(R):
    U, _, Vt = np.linalg.svd(np.asarray(R, dtype=np.float64).reshape(3, 3))
    R = U @ Vt
    rx, ry, rz = R[2,1]-R[1,2], R[0,2]-R[2,0], R[1,0]-R[0,1]
    s = math.sqrt((rx*rx + ry*ry + rz*rz)*0.25)
    c = (R[0,0] + R[1,1] + R[2,2] - 1.0)*0.5
    c = 1.0 if c > 1.0 else (-1.0 if c < -1.0 else c)
    th = math.acos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0,0] + 1)*0.5; rx = math.sqrt(max(t, 0.0))
        t = (R[1,1] + 1)*0.5; ry = math.sqrt(max(t, 0.0))*(-1.0 if R[0,1] < 0 else 1.0)
        t = (R[2,2] + 1)*0.5; rz = math.sqrt(max(t, 0.0))*(-1.0 if R[0,2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1,2] > 0) != (ry*rz > 0)):
            rz = -rz
        th /= math.sqrt(rx*rx + ry*ry + rz*rz)
        return np.array([rx*th, ry*th, rz*th])
    vth = th/(2.0*s)
    return np.array([rx*vth, ry*vth, rz*vth])

def Rodrigues(src):
    a = np.asarray(src, dtype=np.float64)
    if a.size == 9:
        return _mat2vec(a).reshape(3, 1), None
    return _vec2mat(a.ravel()), None

def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs):
    assert distCoeffs is None
    P = np.asarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    rv = np.asarray(rvec, dtype=np.float64)
    R = rv.reshape(3, 3) if rv.size == 9 else _vec2mat(rv.ravel())
    t = np.asarray(tvec, dtype=np.float64).ravel()
    K = np.asarray(cameraMatrix, dtype=np.float64)
    out = np.empty((P.shape[0], 1, 2))
    for i in range(P.shape[0]):
        X = R @ P[i] + t
        z = 1.0/X[2] if X[2] != 0 else 1.0
        out[i, 0, 0] = X[0]*z*K[0,0] + K[0,2]
        out[i, 0, 1] = X[1]*z*K[1,1] + K[1,2]
    return out, None

# --- what src/pipeline.py (and the modules it imports) needs beyond the two calls above ---------------------------
NORM_HAMMING = 6          # default argument in src/features.py:26, evaluated when the class body runs
class DMatch: pass        # annotations in src/pose_estimator.py:7-9, evaluated when the def runs
class KeyPoint: pass

def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2):
    """OpenCV's linear triangulation (DLT) from its published algorithm: per point the 4x4 system
    A = [x1 P1[2] - P1[0]; y1 P1[2] - P1[1]; x2 P2[2] - P2[0]; y2 P2[2] - P2[1]], result = right singular vector of the
    smallest singular value, as a 4 x N array.  OpenCV leaves the vector's sign to its SVD; this stand-in fixes
    w >= 0 (the repository's convention, csrc/ba_triangulate.hpp) -- the reference's '+ 1e-6' on w makes its output
    depend on that choice at the 1e-6 / w level: parity unpinned at the cv2 boundary."""
    P1 = np.asarray(projMatr1, dtype=np.float64); P2 = np.asarray(projMatr2, dtype=np.float64)
    a = np.asarray(projPoints1, dtype=np.float64); b = np.asarray(projPoints2, dtype=np.float64)
    n = a.shape[1]
    out = np.empty((4, n))
    for i in range(n):
        A = np.stack([a[0, i] * P1[2] - P1[0], a[1, i] * P1[2] - P1[1], b[0, i] * P2[2] - P2[0], b[1, i] * P2[2] - P2[1]])
        X = np.linalg.svd(A)[2][3]
        out[:, i] = -X if X[3] < 0 else X
    return out
'''

O3D_STANDIN = '''
class _PC:
    def has_points(self):
        return False
class _G:
    PointCloud = _PC
class _U:
    @staticmethod
    def Vector3dVector(x):
        return x
class _IO:
    @staticmethod
    def write_point_cloud(*a, **k):
        raise RuntimeError("not expected")
geometry = _G()
utility = _U()
io = _IO()
'''

VIS_STANDIN = '''
def plot_and_save_sparsity(*a, **k):
    pass
def plot_and_save_trajectory_2d(*a, **k):
    pass
def plot_and_save_trajectory_3d(*a, **k):
    pass
'''


def load_reference():
    tmp = tempfile.mkdtemp(prefix="ba_standins_")
    for name, src in (("cv2", CV2_STANDIN), ("open3d", O3D_STANDIN), ("visualization", VIS_STANDIN)):
        with open(os.path.join(tmp, name + ".py"), "w") as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, REF_SRC)
    sys.path.insert(0, tmp)
    import bundle_adjuster            # noqa: E402  (the reference file, unmodified)
    import map_structures             # noqa: E402
    return bundle_adjuster, map_structures


class KP:
    def __init__(self, x, y):
        self.pt = (float(np.float32(x)), float(np.float32(y)))


def vec2mat(r):
    import cv2
    return cv2.Rodrigues(np.asarray(r, dtype=np.float64))[0]


def build_scene(ms, seed, n_kf, n_pts, obs_per_pt, K, edge=False):
    """Reference Map with n_kf window keyframes + 1 newest (excluded) keyframe."""
    rng = np.random.default_rng(seed)
    gmap = ms.Map()
    rv = rng.normal(0, 0.05, size=(n_kf, 3))
    rv[0] = 0
    centre = np.stack([np.linspace(0, 1.5, n_kf), 0.1 * rng.normal(size=n_kf), 0.1 * rng.normal(size=n_kf)], 1)
    centre[0] = 0
    if edge:
        rv[1] = 0.0                                    # adjustable camera at theta == 0 exactly
        axis = np.array([0.2, 0.9, 0.1]); axis /= np.linalg.norm(axis)
        rv[2] = axis * (np.pi - 1e-7)                  # theta ~ pi (looks backwards)
        rv[3] = np.array([1e-9, -2e-9, 1.5e-9])        # tiny but > DBL_EPSILON
    Rt = [vec2mat(r) for r in rv]
    tv = [-(Rt[i] @ centre[i]) for i in range(n_kf)]
    X = np.stack([rng.uniform(-2, 3.5, n_pts), rng.uniform(-1.5, 1.5, n_pts), rng.uniform(6, 14, n_pts)], 1)
    if edge:
        X[5] = np.array([0.3, -0.2, -4.0])             # behind every forward-looking camera
    # noisy initial state stored in the map (what BA starts from)
    R0 = []
    for i in range(n_kf):
        d = rng.normal(0, 0.004, 3) if i > 0 else np.zeros(3)
        Rn = vec2mat(rv[i] + d)
        if edge and i == 4:
            Rn = Rn @ (np.eye(3) + 1e-4 * rng.normal(size=(3, 3)))   # slightly non-orthogonal R
        R0.append(Rn)
    t0 = [tv[i] + (rng.normal(0, 0.02, 3) if i > 0 else 0) for i in range(n_kf)]
    if edge:
        R0[1] = np.eye(3)
    X0 = X + rng.normal(0, 0.04, X.shape)
    for j in range(n_pts):
        gmap.add_map_point(ms.MapPoint(id=j, position=X0[j].reshape(3, 1), observations=[], color=np.zeros((3, 1))))
    import cv2
    for i in range(n_kf):
        kps, obs = [], []
        gmap.add_keyframe(ms.Keyframe(id=i, R=R0[i], t=np.asarray(t0[i], dtype=np.float64).reshape(3, 1),
                                      keypoints=kps, descriptors=None, observations=obs, img=None))
    for j in range(n_pts):
        if edge and j == 9:
            cams = [0]                                 # seen only by the fixed keyframe
        else:
            cams = sorted(rng.choice(n_kf, size=min(obs_per_pt, n_kf), replace=False).tolist())
        for i in cams:
            kf = gmap.keyframes[i]
            p, _ = cv2.projectPoints(X[j], rv[i], tv[i], K, None)
            p = p.ravel() + rng.normal(0, 0.5, 2)
            kf.keypoints.append(KP(p[0], p[1]))
            kf.observations.append((j, len(kf.keypoints) - 1))
            gmap.map_points[j].observations.append((i, len(kf.keypoints) - 1))
    if edge:
        kf = gmap.keyframes[3]
        mp_dup = kf.observations[0][0]
        kf.keypoints.append(KP(111.25, 77.5))
        kf.observations.append((mp_dup, len(kf.keypoints) - 1))     # duplicate (kf, mp): last pixel wins
        kf.keypoints.append(KP(5.0, 6.0))
        kf.observations.append((10_000, len(kf.keypoints) - 1))     # map point that does not exist: filtered
    newest = n_kf
    gmap.add_keyframe(ms.Keyframe(id=newest, R=np.eye(3), t=np.zeros((3, 1)), keypoints=[KP(1, 2)],
                                  descriptors=None, observations=[(0, 0)], img=None))
    return gmap


def snapshot(gmap):
    kf_ids = sorted(gmap.keyframes)
    mp_ids = sorted(gmap.map_points)
    return dict(kf_ids=np.array(kf_ids), mp_ids=np.array(mp_ids),
                R=np.array([gmap.keyframes[i].R for i in kf_ids]),
                t=np.array([np.asarray(gmap.keyframes[i].t).reshape(3) for i in kf_ids]),
                X=np.array([np.asarray(gmap.map_points[j].position).reshape(3) for j in mp_ids]))


def dump_map_inputs(gmap):
    """Everything needed to rebuild the same Map on a box without the reference."""
    kf_ids = sorted(gmap.keyframes)
    kf_obs, kf_kps, kf_obs_off, kf_kp_off = [], [], [0], [0]
    for i in kf_ids:
        kf = gmap.keyframes[i]
        kf_obs += [list(o) for o in kf.observations]
        kf_kps += [list(k.pt) for k in kf.keypoints]
        kf_obs_off.append(len(kf_obs))
        kf_kp_off.append(len(kf_kps))
    return dict(in_kf_obs=np.array(kf_obs, dtype=np.int64).reshape(-1, 2),
                in_kf_kps=np.array(kf_kps, dtype=np.float64).reshape(-1, 2),
                in_kf_obs_off=np.array(kf_obs_off), in_kf_kp_off=np.array(kf_kp_off))


def main():
    ba_mod, ms = load_reference()
    from scipy.optimize._numdiff import group_columns
    K = np.array([[912.7820434570312, 0.0, 650.2929077148438],
                  [0.0, 913.0294189453125, 362.7241516113281], [0.0, 0.0, 1.0]])

    # ---- A. cost function + sparsity on regular and edge-case scenes ------------------
    cases = [("cost_seed0", 0, 6, 300, 4, False), ("cost_seed1", 1, 6, 300, 4, False),
             ("cost_seed2", 2, 6, 300, 4, False), ("cost_edge", 7, 6, 60, 3, True)]
    for name, seed, n_kf, n_pts, opp, edge in cases:
        gmap = build_scene(ms, seed, n_kf, n_pts, opp, K, edge)
        ba = ba_mod.BundleAdjuster(K, window_size=n_kf)
        all_ids = sorted(gmap.keyframes.keys())
        local = all_ids[-(ba.window_size + 1):-1]
        fixed, adj = local[0], local[1:]
        mp_ids, observations, kp2d = ba._gather_local_data(gmap, local)
        import cv2
        rv0 = np.array([cv2.Rodrigues(gmap.keyframes[i].R)[0].ravel() for i in adj])
        tv0 = np.array([gmap.keyframes[i].t.ravel() for i in adj])
        X0 = np.array([gmap.map_points[i].position.ravel() for i in mp_ids])
        x0 = np.concatenate([rv0.flatten(), tv0.flatten(), X0.flatten()])
        pose = (gmap.keyframes[fixed].R, gmap.keyframes[fixed].t)
        f0 = ba._cost_function(x0, pose, fixed, adj, mp_ids, observations, kp2d)
        rng = np.random.default_rng(100 + seed)
        x1 = x0 + rng.normal(0, 1e-3, x0.shape)
        f1 = ba._cost_function(x1, pose, fixed, adj, mp_ids, observations, kp2d)
        A = ba._prepare_sparsity_matrix(len(adj), len(mp_ids), adj, mp_ids, observations).tocoo()
        order = np.lexsort((A.col, A.row))
        groups = group_columns(A.tocsr())
        uv_rows = np.array([kp2d[o] for o in observations], dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), K=K, window_size=n_kf,
                            local_kf_ids=np.array(local), fixed_kf_id=fixed, adj_kf_ids=np.array(adj),
                            mp_ids=np.array(mp_ids), observations=np.array(observations, dtype=np.int64),
                            uv_rows=uv_rows, fixed_R=pose[0], fixed_t=np.asarray(pose[1]).reshape(3),
                            x0=x0, f0=f0, x1=x1, f1=f1, sp_rows=A.row[order], sp_cols=A.col[order],
                            sp_shape=np.array(A.shape), n_groups=int(groups.max()) + 1,
                            **{"map_" + k: v for k, v in snapshot(gmap).items()}, **dump_map_inputs(gmap))
        print(name, "obs", len(observations), "n", x0.size, "sse0", float((f0 ** 2).sum()), "groups", int(groups.max()) + 1)

    # ---- B. full run() at reference defaults (log lines, write-back, solver result) ---
    for name, seed, n_kf, n_pts, opp, wsz in [("run_seed0", 0, 6, 120, 4, 6), ("run_seed1", 1, 7, 150, 3, 5),
                                               ("run_global", 2, 5, 100, 3, 6)]:
        gmap = build_scene(ms, 10 + seed, n_kf, n_pts, opp, K, False)
        before = snapshot(gmap)
        inputs = dump_map_inputs(gmap)
        ba = ba_mod.BundleAdjuster(K, window_size=wsz)
        captured = {}
        real_ls = ba_mod.least_squares

        def spy(*a, **k):
            res = real_ls(*a, **k)
            captured["res"] = res
            return res
        ba_mod.least_squares = spy
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ba.run(gmap)
        ba_mod.least_squares = real_ls
        res = captured["res"]
        after = snapshot(gmap)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), K=K, window_size=wsz,
                            log=np.array(buf.getvalue()), res_x=res.x, res_fun=res.fun, res_cost=res.cost,
                            res_nfev=res.nfev, res_status=res.status, res_optimality=res.optimality,
                            **{"before_" + k: v for k, v in before.items()},
                            **{"after_" + k: v for k, v in after.items()}, **inputs)
        print(name, buf.getvalue().strip().splitlines()[-1], "nfev", res.nfev, "status", res.status)

    # ---- C. skip paths -------------------------------------------------------------
    gmap = build_scene(ms, 3, 3, 30, 2, K, False)          # 4 keyframes < window 5
    ba = ba_mod.BundleAdjuster(K)                           # default window_size
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ba.run(gmap)
    log_few = buf.getvalue()
    gmap = build_scene(ms, 4, 1, 30, 1, K, False)          # window of 1 -> no adjustable keyframes
    ba = ba_mod.BundleAdjuster(K, window_size=1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ba.run(gmap)
    log_noadj = buf.getvalue()
    gmap = build_scene(ms, 5, 3, 30, 2, K, False)
    gmap.map_points.clear()                                 # no points survive the membership filter
    ba = ba_mod.BundleAdjuster(K, window_size=3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ba.run(gmap)
    log_nopts = buf.getvalue()
    np.savez_compressed(os.path.join(OUT, "run_skips.npz"), log_few=np.array(log_few), log_noadj=np.array(log_noadj),
                        log_nopts=np.array(log_nopts), default_window=ba_mod.BundleAdjuster(K).window_size)
    print("skips:", repr(log_few), repr(log_noadj), repr(log_nopts))

    # ---- D. converged solutions of the reference residual (scipy, x_scale='jac') -----
    from scipy.optimize import least_squares
    for name, seed, loss in [("conv_linear", 20, "linear"), ("conv_huber", 21, "huber")]:
        gmap = build_scene(ms, seed, 4, 40, 3, K, False)
        ba = ba_mod.BundleAdjuster(K, window_size=4)
        local = sorted(gmap.keyframes.keys())[-5:-1]
        fixed, adj = local[0], local[1:]
        mp_ids, observations, kp2d = ba._gather_local_data(gmap, local)
        import cv2
        x0 = np.concatenate([np.array([cv2.Rodrigues(gmap.keyframes[i].R)[0].ravel() for i in adj]).flatten(),
                             np.array([gmap.keyframes[i].t.ravel() for i in adj]).flatten(),
                             np.array([gmap.map_points[i].position.ravel() for i in mp_ids]).flatten()])
        pose = (gmap.keyframes[fixed].R, gmap.keyframes[fixed].t)
        if loss == "huber":                                  # a few gross outliers so the loss matters
            keys = list(kp2d.keys())
            for kk in keys[::17]:
                kp2d[kk] = (kp2d[kk][0] + 25.0, kp2d[kk][1] - 18.0)
        # dense finite-difference Jacobian + exact trust-region solve: the reference's
        # residual function driven to its minimum (its own LSMR settings stall far from it)
        res = least_squares(ba._cost_function, x0, loss=loss, x_scale='jac', tr_solver='exact',
                            args=(pose, fixed, adj, mp_ids, observations, kp2d), xtol=1e-15, ftol=1e-15,
                            gtol=1e-9, max_nfev=300)
        uv_rows = np.array([kp2d[o] for o in observations], dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), K=K, loss=np.array(loss),
                            local_kf_ids=np.array(local), fixed_kf_id=fixed, adj_kf_ids=np.array(adj),
                            mp_ids=np.array(mp_ids), observations=np.array(observations, dtype=np.int64),
                            uv_rows=uv_rows, fixed_R=pose[0], fixed_t=np.asarray(pose[1]).reshape(3), x0=x0,
                            res_x=res.x, res_fun=res.fun, res_cost=res.cost, res_nfev=res.nfev,
                            res_status=res.status, res_optimality=res.optimality)
        print(name, "nfev", res.nfev, "status", res.status, "cost", res.cost, "opt", res.optimality,
              "rmse", np.sqrt((res.fun ** 2).sum() / len(observations)))


def triangulation_goldens():
    """Section E: the reference's own VisualOdometryPipeline._triangulate_points
    (src/pipeline.py:315-336), imported unmodified, on three two-view scenes -> tests/golden/tri_scenes.npz: inputs, the
    returned (3, kept) array, the kept indices and the printed log line.  cv2.triangulatePoints is the stand-in above."""
    load_reference()
    import pipeline                                   # noqa: E402  (the reference file; its other imports are the reference's own modules)
    K = np.array([[912.7820434570312, 0.0, 650.2929077148438],
                  [0.0, 913.0294189453125, 362.7241516113281], [0.0, 0.0, 1.0]])
    with contextlib.redirect_stdout(io.StringIO()):
        vo = pipeline.VisualOdometryPipeline(K, None, None, None, {})
    import cv2
    out = {"K": K}
    scenes = [("wide", 0, 400, 0.4, 0.5), ("behind", 1, 300, 0.3, 0.3), ("degenerate", 2, 200, 0.05, 0.0)]
    for name, seed, n, base, noise in scenes:
        rng = np.random.default_rng(40 + seed)
        R = cv2.Rodrigues(np.array([0.02, -0.05, 0.01]) * (1.0 if name != "degenerate" else 0.2))[0]
        t = base * np.array([[-1.0], [0.075], [0.125 if name != "behind" else -0.5]])
        X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(5, 20, n)], axis=1)
        if name == "behind":
            X[:25, 2] *= -1.0                          # behind both cameras
            X[25:50] = np.stack([rng.uniform(-0.02, 0.02, 25), rng.uniform(-0.02, 0.02, 25), rng.uniform(0.03, 0.12, 25)], axis=1)
            #  ^ in front of camera 1 (z > 0) but behind camera 2, whose frame is shifted by t_z = -0.15: only the second mask drops them
        if name == "degenerate":
            X[:40] *= 1e7                              # points at "infinity": parallel rays, w ~ 0, the + 1e-6 decides
        x1 = X @ K.T
        x2 = (X @ R.T + t.ravel()) @ K.T
        p1 = x1[:, :2] / x1[:, 2:] + rng.normal(0, noise, (n, 2))
        p2 = x2[:, :2] / x2[:, 2:] + rng.normal(0, noise, (n, 2))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pts, idx = vo._triangulate_points(R, t, p1, p2)
        out.update({f"{name}_R": R, f"{name}_t": t, f"{name}_pts1": p1, f"{name}_pts2": p2, f"{name}_out": pts,
                    f"{name}_idx": idx, f"{name}_log": np.array(buf.getvalue())})
        print(name, "kept", len(idx), "of", n, "|", buf.getvalue().strip())
    with contextlib.redirect_stdout(io.StringIO()):
        empty = vo._triangulate_points(np.eye(3), np.zeros((3, 1)), np.zeros((0, 2)), np.zeros((0, 2)))
    assert empty == (None, None)
    np.savez_compressed(os.path.join(OUT, "tri_scenes.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tri":
        triangulation_goldens()                       # only section E
    else:
        main()
        triangulation_goldens()
