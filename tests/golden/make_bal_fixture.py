#!/usr/bin/env python3
"""Writes tests/golden/tiny_bal.txt: a small BAL-format problem (6 cameras, 40 points) made by this repository's own
generator and writer (no download; the real Ladybug file is not available offline).  Cameras carry distinct
(f, k1, k2); pixels are relative to the image centre; the camera looks down -z (BAL convention)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bundle_adjustment_amd.bal import BALProblem, write_bal            # noqa: E402
from oracle import ba_oracle as o                                     # noqa: E402

rng = np.random.default_rng(7)
nc, npt = 6, 40
rvec = rng.normal(0, 0.05, (nc, 3))
centre = np.stack([np.linspace(0, 2.0, nc), 0.1 * rng.normal(size=nc), 0.1 * rng.normal(size=nc)], 1)
R = o.rodrigues_batch(rvec)
t = -np.einsum("nij,nj->ni", R, centre)
f = 800.0 + 40.0 * rng.normal(size=nc)
k1 = -0.05 + 0.02 * rng.normal(size=nc)
k2 = 0.01 * rng.normal(size=nc)
cams = np.concatenate([rvec, t, f[:, None], k1[:, None], k2[:, None]], axis=1)
pts = np.stack([rng.uniform(-2, 4, npt), rng.uniform(-1.5, 1.5, npt), rng.uniform(-14, -6, npt)], 1)      # in front: -z
cam_idx, pt_idx = [], []
for j in range(npt):
    for c in sorted(rng.choice(nc, size=4, replace=False).tolist()):
        cam_idx.append(c)
        pt_idx.append(j)
cam_idx, pt_idx = np.array(cam_idx, dtype=np.int32), np.array(pt_idx, dtype=np.int32)
proj = -o.bal_residuals(cams, pts, cam_idx, pt_idx, np.zeros((len(cam_idx), 2)))
uv = np.round(proj + rng.normal(0, 0.5, proj.shape), 3)
write_bal(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_bal.txt"), BALProblem(cams, pts, cam_idx, pt_idx, uv))
print("wrote tiny_bal.txt:", nc, "cameras", npt, "points", len(cam_idx), "observations")
