"""Deterministic synthetic bundle-adjustment problems.

Configs (BASELINE.json ``configs``): C1 desk-like (2 cams / 500 pts), C2 (50 / 5k / 30k
obs), C3 headline (1000 / 100k / 1M obs), C5 BAL-like Ladybug topology
(1723 / 156502 / ~678k obs, long-tailed tracks, band-limited visibility).
Intrinsics default to the reference's ``CAMERA_MATRIX`` (``src/main.py:36-40``); the C1
intrinsics are those of ``legacy/local_BA_sparsity_images.py:666-670``.
"""
from __future__ import annotations

import numpy as np

from .map_structures import Keyframe, KeyPoint, Map, MapPoint
from .problem import BAProblem
from .rotations import rvecs_to_matrices

REF_K4 = np.array([912.7820434570312, 913.0294189453125, 650.2929077148438, 362.7241516113281])
DESK_K4 = np.array([431.39865, 431.39865, 429.08605, 235.27142])
IMAGE_WH = (1280.0, 720.0)

CONFIGS = {
    "C1": dict(n_cams=2, n_pts=500, obs_per_pt=2, K4=DESK_K4, image_wh=(848.0, 480.0)),
    "C2": dict(n_cams=50, n_pts=5000, obs_per_pt=6),
    "C3": dict(n_cams=1000, n_pts=100000, obs_per_pt=10),
    # not a BASELINE config: C3 with ten times the points, to see the same kernels away from the launch floor
    "C3x10": dict(n_cams=1000, n_pts=1000000, obs_per_pt=10),
}


def _project(cams, pts, cam_idx, pt_idx, K4):
    R = rvecs_to_matrices(cams[:, :3])
    Xc = np.einsum('nij,nj->ni', R[cam_idx], pts[pt_idx]) + cams[cam_idx, 3:]
    return np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], axis=1), Xc[:, 2]


def _perturb_cameras(rng, rvec, centre, rot_sigma, trans_sigma):
    """Initial guess: rotation vector and camera CENTRE perturbed (camera 0 untouched),
    then t = -R c, so the error does not grow with the distance from the origin."""
    n = rvec.shape[0]
    r0 = rvec.copy()
    c0 = centre.copy()
    r0[1:] += rng.normal(0.0, rot_sigma, size=(n - 1, 3))
    c0[1:] += rng.normal(0.0, trans_sigma, size=(n - 1, 3))
    t0 = -np.einsum('nij,nj->ni', rvecs_to_matrices(r0), c0)
    return np.concatenate([r0, t0], axis=1)


def _observe(rng, cams_true, pts_true, obs_per_pt, K4, image_wh, pixel_sigma, outlier_frac=0.0):
    """Every point is seen by ``obs_per_pt`` distinct cameras chosen uniformly among those that image it inside the frame;
    pixels = projection + N(0, pixel_sigma), rounded to float32.  Returns (cam_idx, pt_idx, uv), camera-major order."""
    n_cams, n_pts = cams_true.shape[0], pts_true.shape[0]
    w, h = image_wh
    # ---- visibility + choice of obs_per_pt distinct cameras per point
    k = min(obs_per_pt, n_cams)
    cam_sel = np.empty((n_pts, k), dtype=np.int32)
    chunk = max(1, min(n_pts, 4_000_000 // max(n_cams, 1)))
    Rm, tm = rvecs_to_matrices(cams_true[:, :3]), cams_true[:, 3:]
    for a in range(0, n_pts, chunk):
        b = min(n_pts, a + chunk)
        X = pts_true[a:b]
        Xc = np.einsum('cij,pj->pci', Rm, X) + tm[None]
        u = K4[0] * Xc[..., 0] / Xc[..., 2] + K4[2]
        v = K4[1] * Xc[..., 1] / Xc[..., 2] + K4[3]
        vis = (Xc[..., 2] > 0.5) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
        key = rng.random((b - a, n_cams))
        key[~vis] += 2.0                         # invisible cameras sort last
        sel = np.argpartition(key, k - 1, axis=1)[:, :k]
        bad = np.take_along_axis(key, sel, axis=1) >= 2.0
        if bad.any():                            # not enough visible cameras: reuse a visible one
            first = np.argmin(key, axis=1)
            sel = np.where(bad, first[:, None], sel)
        cam_sel[a:b] = np.sort(sel, axis=1)
    pt_idx = np.repeat(np.arange(n_pts, dtype=np.int32), k)
    cam_idx = cam_sel.reshape(-1)
    # drop duplicate (cam, pt) pairs created by the fallback above
    pair = cam_idx.astype(np.int64) * n_pts + pt_idx
    _, first = np.unique(pair, return_index=True)
    keep = np.sort(first)
    cam_idx, pt_idx = cam_idx[keep], pt_idx[keep]
    order = np.lexsort((pt_idx, cam_idx))        # camera ascending, then point ascending
    cam_idx, pt_idx = cam_idx[order], pt_idx[order]
    uv, _ = _project(cams_true, pts_true, cam_idx, pt_idx, K4)
    uv = uv + rng.normal(0.0, pixel_sigma, size=uv.shape)
    if outlier_frac > 0:
        nout = int(outlier_frac * uv.shape[0])
        idx = rng.choice(uv.shape[0], size=nout, replace=False)
        uv[idx] += rng.normal(0.0, 30.0, size=(nout, 2))
    uv = uv.astype(np.float32).astype(np.float64)
    return cam_idx, pt_idx, uv


def make_problem(n_cams, n_pts, obs_per_pt, seed=0, K4=REF_K4, image_wh=IMAGE_WH,
                 pixel_sigma=0.5, rot_sigma=0.005, trans_sigma=0.02, point_sigma=0.05,
                 outlier_frac=0.0, return_truth=False):
    """Cameras on a gentle arc looking down +z at a box of points 8-16 m away; every
    point is seen by ``obs_per_pt`` distinct cameras chosen uniformly among those that
    image it inside the frame; pixels get N(0, pixel_sigma) noise and are rounded to
    float32 (keypoint precision); the initial guess is the truth
    perturbed by (rot_sigma, trans_sigma, point_sigma).  Camera 0 is the identity and is
    the fixed camera.  Observation order = camera ascending, then point ascending
    (the reference's gather order for a map whose keyframes list points by id)."""
    rng = np.random.default_rng(seed)
    K4 = np.asarray(K4, dtype=np.float64)
    w, h = image_wh
    # ---- ground truth cameras (world->camera), camera 0 = identity
    s = np.linspace(-1.0, 1.0, n_cams) if n_cams > 1 else np.zeros(1)
    s = s - s[0]
    centre = np.stack([1.5 * s, 0.15 * np.sin(2.0 * s), 0.2 * (1 - np.cos(1.5 * s))], axis=1)
    rvec = rng.normal(0.0, 0.05, size=(n_cams, 3))
    rvec[:, 1] += -0.08 * s                      # pan back towards the scene centre
    rvec[0] = 0.0
    R = rvecs_to_matrices(rvec)
    tvec = -np.einsum('nij,nj->ni', R, centre)
    cams_true = np.concatenate([rvec, tvec], axis=1)
    # ---- ground truth points: box in front of the arc
    span = 1.5 * (s.max() - s.min())
    pts_true = np.stack([rng.uniform(-2.0, 2.0 + span, n_pts) - 0.0,
                         rng.uniform(-1.5, 1.5, n_pts),
                         rng.uniform(8.0, 16.0, n_pts)], axis=1)
    cam_idx, pt_idx, uv = _observe(rng, cams_true, pts_true, obs_per_pt, K4, image_wh, pixel_sigma, outlier_frac)
    # ---- initial guess
    cams0 = _perturb_cameras(rng, rvec, centre, rot_sigma, trans_sigma)
    pts0 = pts_true + rng.normal(0.0, point_sigma, size=pts_true.shape)
    prob = BAProblem(cams0, pts0, cam_idx.astype(np.int32), pt_idx.astype(np.int32), uv, K4.copy(), 0).validate()
    if return_truth:
        return prob, cams_true, pts_true
    return prob


def _arc_cameras(rng, n_cams):
    """Ground-truth cameras of make_problem: a gentle arc looking down +z, camera 0 = identity (same draws, same order)."""
    s = np.linspace(-1.0, 1.0, n_cams) if n_cams > 1 else np.zeros(1)
    s = s - s[0]
    centre = np.stack([1.5 * s, 0.15 * np.sin(2.0 * s), 0.2 * (1 - np.cos(1.5 * s))], axis=1)
    rvec = rng.normal(0.0, 0.05, size=(n_cams, 3))
    rvec[:, 1] += -0.08 * s
    rvec[0] = 0.0
    R = rvecs_to_matrices(rvec)
    tvec = -np.einsum('nij,nj->ni', R, centre)
    return rvec, centre, np.concatenate([rvec, tvec], axis=1), s


def make_weak_shard(n_cams, pts_per_rank, obs_per_pt, seed, rank, K4=REF_K4, image_wh=IMAGE_WH, pixel_sigma=0.5,
                    rot_sigma=0.005, trans_sigma=0.02, point_sigma=0.05):
    """Landmark shard `rank` of a problem that GROWS with the number of ranks (weak scaling: `pts_per_rank` points per
    rank, the same `n_cams` cameras on every rank): cameras -- truth and initial guess -- come from a stream that depends on
    `seed` only, so every rank builds identical ones; the shard's points, visibility choices and pixel noise come from a
    stream of (seed, rank).  The union over ranks is one bundle-adjustment problem of n_ranks * pts_per_rank points in the
    scene of ``make_problem``; no rank ever materialises more than its own shard.  Returns a BAProblem (camera 0 fixed)."""
    K4 = np.asarray(K4, dtype=np.float64)
    rng_c = np.random.default_rng([int(seed), 0])
    rvec, centre, cams_true, s = _arc_cameras(rng_c, n_cams)
    cams0 = _perturb_cameras(rng_c, rvec, centre, rot_sigma, trans_sigma)
    rng_p = np.random.default_rng([int(seed), 1, int(rank)])
    span = 1.5 * (s.max() - s.min())
    pts_true = np.stack([rng_p.uniform(-2.0, 2.0 + span, pts_per_rank), rng_p.uniform(-1.5, 1.5, pts_per_rank),
                         rng_p.uniform(8.0, 16.0, pts_per_rank)], axis=1)
    cam_idx, pt_idx, uv = _observe(rng_p, cams_true, pts_true, obs_per_pt, K4, image_wh, pixel_sigma)
    pts0 = pts_true + rng_p.normal(0.0, point_sigma, size=pts_true.shape)
    return BAProblem(cams0, pts0, cam_idx.astype(np.int32), pt_idx.astype(np.int32), uv, K4.copy(), 0).validate()


def make_config(name, seed=0, **overrides):
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return make_problem(seed=seed, **kw)


def make_bal_like(n_cams=1723, n_pts=156502, n_obs_target=678718, seed=0, K4=REF_K4,
                  image_wh=IMAGE_WH, pixel_sigma=0.5, return_truth=False):
    """BAL 'Ladybug'-like topology with the reference's shared-intrinsics 6-DoF camera:
    cameras along a long street-like path, each point visible only from a band of
    consecutive cameras around where it was first seen, track lengths long-tailed
    (most points 2-4 views, a few tens of views).  The real Ladybug file is not
    available offline; only its counts (1723 / 156502 / 678718) are reproduced."""
    rng = np.random.default_rng(seed)
    K4 = np.asarray(K4, dtype=np.float64)
    w, h = image_wh
    s = np.arange(n_cams) * 0.35                                   # 0.35 m between frames
    centre = np.stack([s, 0.3 * np.sin(s / 25.0), 0.05 * np.cos(s / 40.0)], axis=1)
    rvec = rng.normal(0.0, 0.02, size=(n_cams, 3))
    rvec[0] = 0
    R = rvecs_to_matrices(rvec)
    tvec = -np.einsum('nij,nj->ni', R, centre)
    cams_true = np.concatenate([rvec, tvec], axis=1)
    # long-tailed track lengths with the requested mean
    mean_len = n_obs_target / n_pts
    raw = 2 + rng.pareto(2.2, size=n_pts) * (mean_len - 2) * 1.2
    length = np.clip(np.round(raw).astype(np.int64), 2, min(n_cams, 120))
    start = rng.integers(0, n_cams, size=n_pts)
    start = np.minimum(start, n_cams - length)
    mid = centre[np.minimum(start + length // 2, n_cams - 1)]
    pts_true = mid + np.stack([rng.uniform(-3, 3, n_pts), rng.uniform(-1.5, 1.5, n_pts),
                               rng.uniform(8.0, 20.0, n_pts)], axis=1)
    pt_idx = np.repeat(np.arange(n_pts, dtype=np.int64), length)
    offs = np.arange(pt_idx.shape[0]) - np.repeat(np.cumsum(length) - length, length)
    cam_idx = start[pt_idx] + offs
    uv, z = _project(cams_true, pts_true, cam_idx, pt_idx, K4)
    ok = (z > 0.5) & (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
    # keep at least the two central views of every track so no point is orphaned
    centre_view = np.abs(offs - (length[pt_idx] // 2)) <= 0
    centre_view |= np.abs(offs - (length[pt_idx] // 2 - 1)) <= 0
    keep = ok | centre_view
    cam_idx, pt_idx, uv = cam_idx[keep], pt_idx[keep], uv[keep]
    order = np.lexsort((pt_idx, cam_idx))
    cam_idx, pt_idx, uv = cam_idx[order], pt_idx[order], uv[order]
    uv = (uv + rng.normal(0.0, pixel_sigma, size=uv.shape)).astype(np.float32).astype(np.float64)
    cams0 = _perturb_cameras(rng, rvec, centre, 0.003, 0.02)
    pts0 = pts_true + rng.normal(0.0, 0.05, size=pts_true.shape)
    prob = BAProblem(cams0, pts0, cam_idx.astype(np.int32), pt_idx.astype(np.int32), uv, K4.copy(), 0).validate()
    if return_truth:
        return prob, cams_true, pts_true
    return prob


def bal_project(cams9, pts, cam_idx, pt_idx):
    """Pixels of the BAL camera [rvec | t | f k1 k2] (bal.py): P = R X + t, p = -P.xy / P.z, f (1 + k1 |p|^2 + k2 |p|^4) p.
    Host-side numpy, used to synthesise BAL problems."""
    R = rvecs_to_matrices(cams9[:, :3])
    P = np.einsum('nij,nj->ni', R[cam_idx], pts[pt_idx]) + cams9[cam_idx, 3:6]
    p = -P[:, :2] / P[:, 2:3]
    n2 = (p * p).sum(axis=1, keepdims=True)
    f, k1, k2 = cams9[cam_idx, 6:7], cams9[cam_idx, 7:8], cams9[cam_idx, 8:9]
    return f * (1.0 + n2 * (k1 + k2 * n2)) * p


def make_bal_problem(n_cams=1723, n_pts=156502, n_obs_target=678718, seed=0, pixel_sigma=0.5, focal=900.0):
    """BASELINE config 5 as a BAL problem: the chain topology of ``make_bal_like`` (Ladybug's counts 1723 / 156 502 /
    ~679 k; the real file is not available offline) with the BAL 9-parameter camera -- every camera its own focal length
    (``focal`` +- 2 %) and radial distortion (k1 = -0.03 +- 0.01, k2 = +- 0.003), pixels projected through THAT model
    plus N(0, pixel_sigma) noise and rounded to float32.  Start: poses and points perturbed as in ``make_bal_like``,
    focal lengths off by 0.5 %, k1 off by 0.005, k2 = 0.  Returns a ``bal.BALProblem`` (camera 0 is the natural one to
    hold fixed)."""
    from .bal import BALProblem, from_pinhole
    K4 = np.array([focal, focal, 640.0, 360.0])
    start_pin, cams_true, pts_true = make_bal_like(n_cams, n_pts, n_obs_target, seed=seed, K4=K4, pixel_sigma=0.0,
                                                   return_truth=True)
    rng = np.random.default_rng(seed + 7919)
    truth = from_pinhole(BAProblem(cams_true, pts_true, start_pin.cam_idx, start_pin.pt_idx, start_pin.uv, K4, 0))
    start = from_pinhole(start_pin)
    cams_t = truth.cams.copy()
    cams_t[:, 6] = focal * (1.0 + 0.02 * rng.normal(size=n_cams))
    cams_t[:, 7] = -0.03 + 0.01 * rng.normal(size=n_cams)
    cams_t[:, 8] = 0.003 * rng.normal(size=n_cams)
    uv = bal_project(cams_t, pts_true, truth.cam_idx, truth.pt_idx)
    uv = (uv + rng.normal(0.0, pixel_sigma, size=uv.shape)).astype(np.float32).astype(np.float64)
    cams_0 = start.cams.copy()
    cams_0[:, 6] = cams_t[:, 6] * (1.0 + 0.005 * rng.normal(size=n_cams))
    cams_0[:, 7] = cams_t[:, 7] + 0.005 * rng.normal(size=n_cams)
    cams_0[:, 8] = 0.0
    return BALProblem(cams_0, start.pts.copy(), start.cam_idx.copy(), start.pt_idx.copy(), uv).validate()


def problem_to_map(prob: BAProblem, extra_newest=True):
    """Build a ``Map`` whose window (``all_kf_ids[-(w+1):-1]`` with w = Nc,
    ``src/bundle_adjuster.py:139``) reproduces ``prob``: keyframe i <-> camera i, map
    point j <-> point j, plus one newest keyframe the window excludes."""
    gmap = Map()
    R = rvecs_to_matrices(prob.cams[:, :3])
    per_cam = [[] for _ in range(prob.n_cams)]
    for o in range(prob.n_obs):
        per_cam[int(prob.cam_idx[o])].append(o)
    for j in range(prob.n_pts):
        gmap.add_map_point(MapPoint(id=j, position=prob.pts[j].reshape(3, 1).copy(), observations=[],
                                    color=np.full((3, 1), 0.5)))
    for i in range(prob.n_cams):
        kps, obs = [], []
        for n, o in enumerate(per_cam[i]):
            kps.append(KeyPoint(pt=(float(prob.uv[o, 0]), float(prob.uv[o, 1]))))
            obs.append((int(prob.pt_idx[o]), n))
            gmap.map_points[int(prob.pt_idx[o])].observations.append((i, n))
        gmap.add_keyframe(Keyframe(id=i, R=R[i].copy(), t=prob.cams[i, 3:].reshape(3, 1).copy(),
                                   keypoints=kps, descriptors=None, observations=obs, img=None))
    if extra_newest:
        i = prob.n_cams
        gmap.add_keyframe(Keyframe(id=i, R=np.eye(3), t=np.zeros((3, 1)), keypoints=[], descriptors=None,
                                   observations=[], img=None))
    return gmap
