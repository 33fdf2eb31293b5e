"""Shared helpers for the tests: load golden fixtures, rebuild maps from them."""
import os

import numpy as np

from bundle_adjustment_amd.map_structures import Keyframe, KeyPoint, Map, MapPoint

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_cost_case(g):
    """-> args of the reference's _cost_function as python objects."""
    obs = [(int(a), int(b)) for a, b in g["observations"]]
    kp2d = {o: (float(u), float(v)) for o, (u, v) in zip(obs, g["uv_rows"])}
    pose = (g["fixed_R"], g["fixed_t"].reshape(3, 1))
    return dict(fixed_kf_pose=pose, fixed_kf_id=int(g["fixed_kf_id"]),
                adjustable_kf_ids=[int(i) for i in g["adj_kf_ids"]],
                map_point_ids=[int(i) for i in g["mp_ids"]], observations=obs, keypoints_2d=kp2d)


def golden_flat_problem(g, x=None):
    """Flat SoA problem (cams incl. the fixed one at index 0) from a cost/conv golden."""
    from oracle import ba_oracle as o
    from bundle_adjustment_amd.problem import BAProblem
    x = g["x0"] if x is None else x
    adj = [int(i) for i in g["adj_kf_ids"]]
    na = len(adj)
    mp = [int(i) for i in g["mp_ids"]]
    cams = np.zeros((na + 1, 6))
    cams[0, :3] = o.rodrigues_to_vec(g["fixed_R"])
    cams[0, 3:] = g["fixed_t"]
    cams[1:, :3] = x[:3 * na].reshape(na, 3)
    cams[1:, 3:] = x[3 * na:6 * na].reshape(na, 3)
    pts = x[6 * na:].reshape(len(mp), 3).copy()
    kf_index = {int(g["fixed_kf_id"]): 0}
    kf_index.update({k: i + 1 for i, k in enumerate(adj)})
    mp_index = {m: i for i, m in enumerate(mp)}
    cam_idx = np.array([kf_index[int(a)] for a, _ in g["observations"]], dtype=np.int32)
    pt_idx = np.array([mp_index[int(b)] for _, b in g["observations"]], dtype=np.int32)
    K = g["K"]
    K4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
    return BAProblem(cams, pts, cam_idx, pt_idx, g["uv_rows"].copy(), K4, 0)


def rebuild_map(g, prefix="before_"):
    """Map in this package's own classes from a run_* golden (state `prefix`)."""
    gmap = Map()
    kf_ids = [int(i) for i in g[prefix + "kf_ids"]]
    for j, X in zip(g[prefix + "mp_ids"], g[prefix + "X"]):
        gmap.add_map_point(MapPoint(id=int(j), position=X.reshape(3, 1).copy(), observations=[],
                                    color=np.zeros((3, 1))))
    for n, i in enumerate(kf_ids):
        o0, o1 = int(g["in_kf_obs_off"][n]), int(g["in_kf_obs_off"][n + 1])
        k0, k1 = int(g["in_kf_kp_off"][n]), int(g["in_kf_kp_off"][n + 1])
        kps = [KeyPoint(pt=(float(u), float(v))) for u, v in g["in_kf_kps"][k0:k1]]
        obs = [(int(a), int(b)) for a, b in g["in_kf_obs"][o0:o1]]
        gmap.add_keyframe(Keyframe(id=i, R=g[prefix + "R"][n].copy(), t=g[prefix + "t"][n].reshape(3, 1).copy(),
                                   keypoints=kps, descriptors=None, observations=obs, img=None))
    return gmap
