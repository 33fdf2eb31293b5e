#!/usr/bin/env python3
"""A rank's share of a multi-GPU job, timed on ONE GPU: a communicator of one rank (BA_COMM_FORCE=1 semantics: every
all-reduce of the multi-rank control flow is issued) over a shard of C3 -- all 1000 cameras, 100000 / N points.

    python tools/shard_times.py N [N ...]          # BA_ONE_PART=0/1 forces the partition layout
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BA_COMM_FORCE"] = "1"
from bundle_adjustment_amd import hip_backend as hb             # noqa: E402
from bundle_adjustment_amd.synthetic import make_problem         # noqa: E402

for n in (int(a) for a in sys.argv[1:]):
    p = make_problem(1000, 100000 // n, 10, seed=0)
    s = hb.Solver(0)
    s.comm_init(0, 1, hb.comm_unique_id())
    s.set_problem(p)
    kw = dict(loss="huber", max_iters=20, ftol=0.0, xtol=0.0, gtol=1e-300, pcg_tol=0.1, pcg_max_iters=200)
    ts = []
    for r in range(7):
        s.set_params(p.cams, p.pts)
        t = time.perf_counter(); out = s.solve(**kw); ts.append(time.perf_counter() - t)
    t = float(np.median(ts[1:]))
    print(f"shard 1/{n}: {p.n_pts} pts {p.n_obs} obs: {out['iterations'] / t:8.1f} LM it/s, {out['pcg_iterations'] / out['iterations']:5.2f} PCG/LM, "
          f"{t / out['iterations'] * 1e6:7.1f} us per LM iteration")
    s.close()
