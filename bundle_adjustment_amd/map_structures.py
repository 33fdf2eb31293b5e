"""Map data model consumed and produced by ``BundleAdjuster.run``.

Mirrors the reference's ``src/map_structures.py:7-54`` field for field (``MapPoint``,
``Keyframe``, ``Map`` with its two dicts and id counters) so a map built by the
reference's pipeline can be handed to this package unchanged and vice versa.  Bundle
adjustment reads ``(R, t)`` as world->camera: ``Xc = R @ X + t``
(``src/bundle_adjuster.py:58-67``).

Differences, all outside the solve path: no open3d import; ``get_pcd`` returns a small
numpy-backed stand-in (``has_points()``, ``points``, ``colors``) and is only a
convenience for the optional snapshot step of ``run``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class KeyPoint:
    """Stand-in for ``cv2.KeyPoint``: only ``.pt`` is read by bundle adjustment
    (``src/bundle_adjuster.py:216``)."""
    pt: tuple


@dataclass
class MapPoint:
    id: int
    position: np.ndarray          # (3,1)
    observations: list            # [(keyframe_id, keypoint_index)]
    color: np.ndarray = field(default_factory=lambda: np.zeros((3, 1)))


@dataclass
class Keyframe:
    id: int
    R: np.ndarray                 # (3,3) world -> camera
    t: np.ndarray                 # (3,1)
    keypoints: list               # objects with .pt = (x, y)
    descriptors: np.ndarray = None
    observations: list = field(default_factory=list)   # [(map_point_id, keypoint_index)]
    img: np.ndarray = None


class PointCloud:
    """Minimal numpy point cloud used when open3d is not installed."""

    def __init__(self, points=None, colors=None):
        self.points = np.zeros((0, 3)) if points is None else points
        self.colors = np.zeros((0, 3)) if colors is None else colors

    def has_points(self):
        return self.points.shape[0] > 0


class Map:
    def __init__(self):
        self.keyframes = {}
        self.map_points = {}
        self.next_keyframe_id = 0
        self.next_map_point_id = 0

    def add_keyframe(self, keyframe: Keyframe):
        if keyframe.id in self.keyframes:
            raise ValueError(f"Keyframe with ID {keyframe.id} already exists.")
        self.keyframes[keyframe.id] = keyframe
        self.next_keyframe_id += 1

    def add_map_point(self, map_point: MapPoint):
        if map_point.id in self.map_points:
            raise ValueError(f"MapPoint with ID {map_point.id} already exists.")
        self.map_points[map_point.id] = map_point
        self.next_map_point_id += 1

    def get_pcd(self):
        if not self.map_points:
            return PointCloud()
        pts = np.array([p.position for p in self.map_points.values()], dtype=np.float64).reshape(-1, 3)
        cols = np.array([p.color for p in self.map_points.values()], dtype=np.float64).reshape(-1, 3)
        return PointCloud(pts, cols)
