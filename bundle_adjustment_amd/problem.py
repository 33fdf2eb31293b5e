"""Flat (structure-of-arrays) form of a bundle-adjustment window and the walk that
builds it from a ``Map``.

``BAProblem`` is what crosses the C-ABI (``include/ba_hip.h``): cameras ``(Nc,6)`` =
``[rvec | tvec]`` (world->camera), points ``(Np,3)``, observations as
``cam_idx[int32]``, ``pt_idx[int32]``, ``uv[f64,2]`` in the reference's residual row
order, intrinsics ``(fx, fy, cx, cy)`` and the index of the fixed camera.

``gather_window`` follows ``BundleAdjuster._gather_local_data``
(``src/bundle_adjuster.py:195-218``): keyframes in the given order, each keyframe's
``observations`` in insertion order, observations kept when the map point still exists,
map-point ids sorted ascending, and for a repeated ``(keyframe, map point)`` pair the
LAST keypoint wins for both rows (the reference stores pixels in a dict keyed by the
pair).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .rotations import matrices_to_rvecs


@dataclass
class BAProblem:
    cams: np.ndarray        # (Nc,6) float64  [rvec | tvec]
    pts: np.ndarray         # (Np,3) float64
    cam_idx: np.ndarray     # (Nobs,) int32
    pt_idx: np.ndarray      # (Nobs,) int32
    uv: np.ndarray          # (Nobs,2) float64
    K4: np.ndarray          # (4,) fx, fy, cx, cy
    fixed_cam: int = 0      # index into cams, -1 = none fixed

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
        """Shape / range checks done on the host before anything reaches a kernel."""
        nc, npt, nobs = self.n_cams, self.n_pts, self.n_obs
        if self.cams.shape != (nc, 6) or self.pts.shape != (npt, 3):
            raise ValueError("cams must be (Nc,6) and pts (Np,3)")
        if self.pt_idx.shape != (nobs,) or self.uv.shape != (nobs, 2):
            raise ValueError("observation arrays disagree in length")
        if nobs:
            if self.cam_idx.min() < 0 or self.cam_idx.max() >= nc:
                raise ValueError("cam_idx out of range")
            if self.pt_idx.min() < 0 or self.pt_idx.max() >= npt:
                raise ValueError("pt_idx out of range")
        if not (-1 <= self.fixed_cam < nc):
            raise ValueError("fixed_cam out of range")
        return self


def gather_window(gmap, local_kf_ids):
    """-> (sorted map-point ids, observations [(kf_id, mp_id)], keypoints_2d dict)."""
    mp_ids = set()
    observations = []
    keypoints_2d = {}
    have = gmap.map_points
    for kf_id in local_kf_ids:
        kf = gmap.keyframes[kf_id]
        kps = kf.keypoints
        for mp_id, kp_idx in kf.observations:
            if mp_id in have:
                mp_ids.add(mp_id)
                observations.append((kf_id, mp_id))
                keypoints_2d[(kf_id, mp_id)] = kps[kp_idx].pt
    return sorted(mp_ids), observations, keypoints_2d


def flatten_window(gmap, local_kf_ids, mp_ids, observations, keypoints_2d, camera_matrix):
    """Pack a gathered window into a ``BAProblem``.  Camera 0 of the result is
    ``local_kf_ids[0]`` (the fixed keyframe, ``src/bundle_adjuster.py:141``)."""
    kf_index = {kf: i for i, kf in enumerate(local_kf_ids)}
    mp_index = {mp: i for i, mp in enumerate(mp_ids)}
    nobs = len(observations)
    cam_idx = np.fromiter((kf_index[k] for k, _ in observations), dtype=np.int32, count=nobs)
    pt_idx = np.fromiter((mp_index[m] for _, m in observations), dtype=np.int32, count=nobs)
    uv = np.array([keypoints_2d[o] for o in observations], dtype=np.float64).reshape(nobs, 2)
    Rs = np.array([gmap.keyframes[k].R for k in local_kf_ids], dtype=np.float64)
    ts = np.array([np.asarray(gmap.keyframes[k].t, dtype=np.float64).ravel() for k in local_kf_ids])
    cams = np.concatenate([matrices_to_rvecs(Rs), ts], axis=1)
    pts = np.array([np.asarray(gmap.map_points[m].position, dtype=np.float64).ravel() for m in mp_ids])
    K = np.asarray(camera_matrix, dtype=np.float64)
    K4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
    return BAProblem(cams, pts.reshape(-1, 3), cam_idx, pt_idx, uv, K4, fixed_cam=0).validate()


def _window_cameras(gmap, local_kf_ids, camera_matrix):
    Rs = np.array([gmap.keyframes[k].R for k in local_kf_ids], dtype=np.float64)
    ts = np.array([np.asarray(gmap.keyframes[k].t, dtype=np.float64).ravel() for k in local_kf_ids])
    cams = np.concatenate([matrices_to_rvecs(Rs), ts], axis=1)
    K = np.asarray(camera_matrix, dtype=np.float64)
    return cams, np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])


def flatten_map_window_numpy(gmap, local_kf_ids, camera_matrix):
    """Array-level numpy form of ``flatten_map_window`` (one pass per keyframe, no per-observation
    tuples or dict entries).  Kept as the plain statement the native walk is tested against."""
    have = gmap.map_points
    have_ids = np.fromiter(have.keys(), dtype=np.int64, count=len(have))
    cam_parts, mp_parts, uv_parts = [], [], []
    for ci, kf_id in enumerate(local_kf_ids):
        kf = gmap.keyframes[kf_id]
        if not kf.observations:
            continue
        ob = np.asarray(kf.observations, dtype=np.int64).reshape(-1, 2)
        keep = np.isin(ob[:, 0], have_ids)
        if not keep.all():
            ob = ob[keep]
        if ob.shape[0] == 0:
            continue
        mp, kp_idx = ob[:, 0], ob[:, 1]
        # a repeated map point inside one keyframe: every row takes the LAST keypoint listed for it
        uniq, inv = np.unique(mp, return_inverse=True)
        if uniq.shape[0] != mp.shape[0]:
            last = np.zeros(uniq.shape[0], dtype=np.int64)
            last[inv] = np.arange(mp.shape[0])            # later rows overwrite earlier ones
            kp_idx = kp_idx[last[inv]]
        kps = kf.keypoints
        used, uinv = np.unique(kp_idx, return_inverse=True)
        pix = np.array([kps[i].pt for i in used.tolist()], dtype=np.float64).reshape(-1, 2)
        cam_parts.append(np.full(mp.shape[0], ci, dtype=np.int32))
        mp_parts.append(mp)
        uv_parts.append(pix[uinv])
    if not mp_parts:
        return None, []
    mp_all = np.concatenate(mp_parts)
    mp_ids, pt_idx = np.unique(mp_all, return_inverse=True)
    cams, K4 = _window_cameras(gmap, local_kf_ids, camera_matrix)
    pts = np.array([np.asarray(have[int(m)].position, dtype=np.float64).ravel() for m in mp_ids.tolist()]).reshape(-1, 3)
    prob = BAProblem(cams, pts, np.concatenate(cam_parts), pt_idx.astype(np.int32), np.concatenate(uv_parts), K4,
                     fixed_cam=0).validate()
    return prob, [int(m) for m in mp_ids.tolist()]


def flatten_map_window(gmap, local_kf_ids, camera_matrix):
    """``flatten_map_window_ids`` with the sorted map-point ids as a list (the reference's ``local_map_point_ids``)."""
    prob, mp_ids = flatten_map_window_ids(gmap, local_kf_ids, camera_matrix)
    return prob, mp_ids.tolist()


def flatten_map_window_ids(gmap, local_kf_ids, camera_matrix):
    """``gather_window`` + ``flatten_window`` in one pass: same ``BAProblem`` (same row order, same
    last-pixel-wins rule for a repeated ``(keyframe, map point)`` pair) plus the sorted map-point
    ids (int64 array).  The walk over the Map objects runs in the ``_mapwalk`` C extension (``csrc/mapwalk.c``,
    built by ``__graft_entry__.build()``); what is left here is array work on its outputs.  Used by
    ``BundleAdjuster.run``; the tuple/dict form stays available through ``_gather_local_data``."""
    from . import _mapwalk
    keyframes, have = gmap.keyframes, gmap.map_points
    cap = sum(len(keyframes[k].observations) for k in local_kf_ids)
    if cap == 0:
        return None, np.empty(0, dtype=np.int64)
    cam_idx = np.empty(cap, dtype=np.int32)
    first_seen = np.empty(cap, dtype=np.int64)
    uv = np.empty((cap, 2), dtype=np.float64)
    distinct = np.empty(cap, dtype=np.int64)
    nobs, npts = _mapwalk.walk_window(keyframes, have, list(local_kf_ids), cam_idx, first_seen, uv, distinct)
    if nobs == 0:
        return None, np.empty(0, dtype=np.int64)
    distinct = distinct[:npts]
    order = np.argsort(distinct, kind="stable")          # map-point ids ascending (:210 sorted(...))
    rank = np.empty(npts, dtype=np.int32)
    rank[order] = np.arange(npts, dtype=np.int32)
    mp_ids = distinct[order]
    pts = np.empty((npts, 3), dtype=np.float64)
    _mapwalk.gather_positions(have, mp_ids, pts)
    cams, K4 = _window_cameras(gmap, local_kf_ids, camera_matrix)
    # (indices come out of the walk itself; hip_backend.set_problem and ba_set_problem both check ranges again)
    prob = BAProblem(cams, pts, cam_idx[:nobs], rank[first_seen[:nobs]], uv[:nobs], K4, fixed_cam=0)
    return prob, mp_ids


def shard_by_landmark(problem: BAProblem, n_shards: int):
    """Split the points into ``n_shards`` contiguous index ranges balanced by
    observation count.  Returns ``[(p_begin, p_end)]``; shard g owns those points and
    every observation of them; cameras are replicated.  No shard is
    empty when ``n_pts >= n_shards``; with fewer points than shards the trailing shards are
    empty (the library copes: an empty rank still joins every collective)."""
    npt = problem.n_pts
    counts = np.bincount(problem.pt_idx, minlength=npt).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = cum[-1]
    bounds = [0]
    for g in range(1, n_shards):
        target = total * g / n_shards
        b = int(np.searchsorted(cum, target, side='left'))
        bounds.append(min(max(b, bounds[-1]), npt))
    bounds.append(npt)
    if npt >= n_shards:
        # never an empty shard while there are at least as many points as shards (a few very long tracks can pull
        # two cuts onto the same point): every shard keeps one point or more
        for g in range(1, n_shards):
            bounds[g] = max(bounds[g], bounds[g - 1] + 1)
        for g in range(n_shards - 1, 0, -1):
            bounds[g] = min(bounds[g], npt - (n_shards - g))
    return [(bounds[g], bounds[g + 1]) for g in range(n_shards)]


def extract_shard(problem: BAProblem, p_begin: int, p_end: int):
    """Sub-problem holding points [p_begin, p_end) (re-indexed from 0), their
    observations in the original relative order, and ALL cameras.  Also returns the
    positions of those observations in the full list."""
    sel = np.nonzero((problem.pt_idx >= p_begin) & (problem.pt_idx < p_end))[0]
    sub = BAProblem(problem.cams.copy(), problem.pts[p_begin:p_end].copy(),
                    problem.cam_idx[sel].copy(), (problem.pt_idx[sel] - p_begin).astype(np.int32),
                    problem.uv[sel].copy(), problem.K4.copy(), problem.fixed_cam)
    return sub, sel


class WindowCache:
    """What ``BundleAdjuster.run`` keeps between consecutive calls so that an UNCHANGED window is not walked, re-sorted and
    re-uploaded again.  The reference calls ``run`` after every new keyframe (``src/pipeline.py:99``) and once more for the
    final global BA (``src/main.py:83-86``).

    One level, validated on every use (never trusted blindly): the whole window -- same keyframe objects, the same
    observation-list objects with the same lengths, and every landmark the cached window lists still in
    ``gmap.map_points`` -> the observation structure (camera index, landmark index, pixels, sorted landmark ids) is reused
    as it stands; only poses and positions are read again, and the caller may keep the solver's uploaded problem and send
    parameters only (a repeated run on the same window, e.g. the global BA after the last sliding-window one: 188 ms -> 10 ms
    at 1000 keyframes / 1 M observations).  A window that moved is walked afresh by the native walk (``csrc/mapwalk.c``,
    ~0.1 us per observation).  (Round 2 also cached the rows of single keyframes, so that a sliding window walked only the
    keyframe that entered; measured on the reference's own 5-keyframe use that bought nothing -- the per-keyframe
    validation and the numpy merge cost what the walk of four small keyframes costs -- and it is gone.)

    Assumption, stated: observation lists are append-only (what the reference does); an element replaced in place
    without changing the length is not noticed.  The cache holds references to the list objects it has seen, so an
    ``id()`` can never be recycled behind its back.  ``BundleAdjuster(..., reuse_window=False)`` turns it off; windows with
    fewer than ``min_obs`` observation-list entries (``BundleAdjuster(..., reuse_min_obs=20000)``) bypass it: walking a
    sliding window of five keyframes costs 0.1 ms, less than three times the cache's own bookkeeping, and such a window
    moves on every call.
    """

    def __init__(self, min_obs=20000):
        self.window = None          # dict: ids, kfs, lists, lens, n_have, all_present, cam_idx, pt_idx, uv, mp_ids, token
        self.tokens = 0
        self.hits = dict(window=0, walked=0)
        # windows with fewer observation-list entries than this are simply walked every time (a sliding window of five
        # keyframes: the walk is 0.1 ms, the cache's bookkeeping and validation would be a third of that, for a window
        # that moves on every call anyway)
        self.min_obs = int(min_obs)

    def flatten(self, gmap, local_kf_ids, camera_matrix):
        """-> (BAProblem or None, sorted landmark ids as int64 array, structure token).  The token changes whenever the
        observation structure (indices, pixels) differs from the previous call's; equal tokens = same structure."""
        from . import _mapwalk
        keyframes, have = gmap.keyframes, gmap.map_points
        kfs = [keyframes[k] for k in local_kf_ids]
        lists = [kf.observations for kf in kfs]
        lens = [len(o) for o in lists]
        if sum(lens) < self.min_obs:
            prob, mp_ids = flatten_map_window_ids(gmap, local_kf_ids, camera_matrix)
            self.hits["walked"] += len(kfs)
            self.window = None
            self.tokens += 1
            return prob, mp_ids, (self.tokens if prob is not None else -1)
        w = self.window
        reuse = (w is not None and len(w["kfs"]) == len(kfs) and w["ids"] == list(local_kf_ids)
                 and all(a is b for a, b in zip(w["kfs"], kfs)) and all(a is b for a, b in zip(w["lists"], lists))
                 and w["lens"] == lens and (w["all_present"] or w["n_have"] == len(have))
                 and _mapwalk.count_present(have, w["mp_ids"]) == w["mp_ids"].shape[0])
        if not reuse:
            prob, mp_ids = flatten_map_window_ids(gmap, local_kf_ids, camera_matrix)
            self.hits["walked"] += len(kfs)
            if prob is None:
                self.window = None
                return None, np.empty(0, dtype=np.int64), -1
            self.tokens += 1
            self.window = dict(ids=list(local_kf_ids), kfs=kfs, lists=lists, lens=lens, n_have=len(have),
                               all_present=(prob.n_obs == sum(lens)), cam_idx=prob.cam_idx, pt_idx=prob.pt_idx, uv=prob.uv,
                               mp_ids=mp_ids, token=self.tokens)
            return prob, self.window["mp_ids"], self.tokens
        self.hits["window"] += 1
        mp_ids = w["mp_ids"]
        pts = np.empty((mp_ids.shape[0], 3), dtype=np.float64)
        _mapwalk.gather_positions(have, mp_ids, pts)
        cams, K4 = _window_cameras(gmap, local_kf_ids, camera_matrix)
        return BAProblem(cams, pts, w["cam_idx"], w["pt_idx"], w["uv"], K4, fixed_cam=0), mp_ids, w["token"]
