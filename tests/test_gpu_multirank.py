"""GPU: two ranks sharing ONE GPU through the shared-memory test transport (BA_COMM=shm; RCCL
refuses two ranks on one device, and only a 1-GPU box is available for tests).  Exercises the
whole multi-rank control flow of ba_solve -- landmark shards, folded partials, all-reduces at
every exchange point, host polling of PCG verdicts -- and compares with the single-rank solve."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.problem import extract_shard, shard_by_landmark
from bundle_adjustment_amd.synthetic import make_problem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
p = make_problem(14, 1500, 5, seed=11, outlier_frac=0.02)
b, e = shard_by_landmark(p, world)[rank]
sub, _ = extract_shard(p, b, e)
s = hip_backend.Solver(0)
uid = [hip_backend.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
s.comm_init(rank, world, uid[0])
s.set_problem(sub)
out = s.solve(loss="huber", max_iters=25, ftol=1e-13, xtol=1e-13, gtol=1e-12, pcg_tol=1e-3)
cams, pts = s.get_params()
np.save(os.path.join(%(out)r, f"cams_{rank}.npy"), cams)
np.save(os.path.join(%(out)r, f"pts_{rank}.npy"), pts)
json.dump(out, open(os.path.join(%(out)r, f"out_{rank}.json"), "w"))
s.close()
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_match_single_rank(tmp_path):
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.problem import shard_by_landmark
    from bundle_adjustment_amd.synthetic import make_problem
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, BA_COMM="shm")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    p = make_problem(14, 1500, 5, seed=11, outlier_frac=0.02)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        ref = s.solve(loss="huber", max_iters=25, ftol=1e-13, xtol=1e-13, gtol=1e-12, pcg_tol=1e-3)
        cams_ref, pts_ref = s.get_params()
    outs = [json.load(open(tmp_path / f"out_{k}.json")) for k in range(2)]
    # every rank reports the same global costs / iteration counts
    for key in ("iterations", "accepted", "pcg_iterations", "initial_sse", "final_sse", "final_cost"):
        assert outs[0][key] == outs[1][key], key
    assert abs(outs[0]["initial_sse"] - ref["initial_sse"]) <= 1e-10 * ref["initial_sse"]
    assert abs(outs[0]["final_cost"] - ref["final_cost"]) <= 1e-9 * ref["final_cost"]
    cams0, cams1 = np.load(tmp_path / "cams_0.npy"), np.load(tmp_path / "cams_1.npy")
    assert np.array_equal(cams0, cams1)                     # replicated cameras stay bitwise identical
    assert np.abs(cams0 - cams_ref).max() <= 1e-6
    ranges = shard_by_landmark(p, 2)
    pts = np.concatenate([np.load(tmp_path / f"pts_{k}.npy") for k in range(2)])
    assert pts.shape == pts_ref.shape and ranges[1][1] == p.n_pts
    assert np.abs(pts - pts_ref).max() <= 1e-5
