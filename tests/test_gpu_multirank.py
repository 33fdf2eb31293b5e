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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)

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
# second solve from the start: plain Jacobi blocks (the damped system's message then carries no Schur-Jacobi blocks) and a
# gradient tolerance that ends the run -- every rank's max |bp| travels in that message's tail
s.set_params(sub.cams, sub.pts)
out2 = s.solve(loss="huber", max_iters=25, ftol=0.0, xtol=0.0, gtol=float(os.environ["TEST_GTOL"]), pcg_tol=1e-3, preconditioner="jacobi")
json.dump(out2, open(os.path.join(%(out)r, f"out2_{rank}.json"), "w"))
json.dump(s.stats(), open(os.path.join(%(out)r, f"stats_{rank}.json"), "w"))
s.close()
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("one_part,ipc", [("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")])
def test_two_ranks_on_one_gpu_match_single_rank(tmp_path, one_part, ipc):
    """one_part = 1: every camera's shard-local observations in partition 0 (BA_ONE_PART, an experiment switch of
    ba_set_problem): partial sums come out folded, no fold kernel runs ahead of the all-reduces.
    ipc = 1 (BA_IPC): the per-PCG-iteration exchange of the reduced camera system's product does not go through the
    transport's all-reduce but through IPC-mapped peer buffers -- every rank's fold kernel stores its share and a sequence
    flag into the other rank's receive buffer (hipIpcGetMemHandle / hipIpcOpenMemHandle between the two processes on this
    one GPU), k_pcg_step waits for the flags and adds the slots in rank order.  Same sums in the same order as the
    host-staged transport: the results must be IDENTICAL to the ipc = 0 run's, bit for bit (checked through the
    single-rank reference both are compared with, and by the exchange counter)."""
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.problem import shard_by_landmark
    from bundle_adjustment_amd.synthetic import make_problem
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    # a gradient tolerance that ends the single-rank run after a few accepted steps (found here, handed to the workers)
    p = make_problem(14, 1500, 5, seed=11, outlier_frac=0.02)
    gkw = dict(loss="huber", max_iters=25, ftol=0.0, xtol=0.0, pcg_tol=1e-3, preconditioner="jacobi")
    ref2 = None
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        for gtol in (1e4, 1e3, 1e2, 1e1, 1.0, 1e-1, 1e-2, 1e-3, 1e-4):
            s.set_params(p.cams, p.pts)
            cand = s.solve(gtol=gtol, **gkw)
            if cand["status"] == 3 and cand["iterations"] >= 2:
                ref2 = cand
                break
    assert ref2 is not None and ref2["iterations"] < 25
    env = dict(os.environ, BA_COMM="shm", BA_ONE_PART=one_part, BA_IPC=ipc, TEST_GTOL=repr(gtol))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        ref = s.solve(loss="huber", max_iters=25, ftol=1e-13, xtol=1e-13, gtol=1e-12, pcg_tol=1e-3)
        cams_ref, pts_ref = s.get_params()
    outs = [json.load(open(tmp_path / f"out_{k}.json")) for k in range(2)]
    stats = [json.load(open(tmp_path / f"stats_{k}.json")) for k in range(2)]
    for st in stats:                                         # one exchange per PCG iteration went through the peer buffers, or none
        assert (st["ipc_exchanges"] >= outs[0]["pcg_iterations"]) if ipc == "1" else (st["ipc_exchanges"] == 0), st
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
    # the gradient-tolerance stop: same verdict at the same iteration on both ranks and on one rank
    outs2 = [json.load(open(tmp_path / f"out2_{k}.json")) for k in range(2)]
    for o in outs2:
        assert o["status"] == 3 and o["iterations"] == ref2["iterations"]
        assert abs(o["final_cost"] - ref2["final_cost"]) <= 1e-9 * ref2["final_cost"]


RUN_WORKER = r"""
import io, os, sys
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from bundle_adjustment_amd import BundleAdjuster, hip_backend
from bundle_adjustment_amd.synthetic import make_problem, problem_to_map
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
p = make_problem(12, 1200, 5, seed=17)
gmap = problem_to_map(p)
K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
uid = [hip_backend.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
ba = BundleAdjuster(K, window_size=p.n_cams, comm=(rank, world, uid[0]), ftol=1e-12, xtol=1e-12, pcg_tol=1e-3)
buf = io.StringIO()
with redirect_stdout(buf):
    ba.run(gmap)
ids = sorted(gmap.map_points)
np.save(os.path.join(%(out)r, f"run_pts_{rank}.npy"), np.array([gmap.map_points[i].position.ravel() for i in ids]))
np.save(os.path.join(%(out)r, f"run_R_{rank}.npy"), np.array([gmap.keyframes[k].R for k in sorted(gmap.keyframes)]))
open(os.path.join(%(out)r, f"run_log_{rank}.txt"), "w").write(buf.getvalue())
ba.close()
dist.barrier()
dist.destroy_process_group()
"""


def test_spmd_run_on_two_ranks_leaves_the_same_map_everywhere(tmp_path):
    """BundleAdjuster.run with comm=(rank, world, id): each rank solves its landmark block, the points
    of all blocks are gathered (ba_allgather_points) and both ranks write back the same map as a
    single-rank run (to the rounding of the re-ordered sums)."""
    import io
    from contextlib import redirect_stdout
    from bundle_adjustment_amd import BundleAdjuster
    from bundle_adjustment_amd.synthetic import make_problem, problem_to_map
    script = tmp_path / "run_worker.py"
    script.write_text(RUN_WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, BA_COMM="shm")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    p = make_problem(12, 1200, 5, seed=17)
    gmap = problem_to_map(p)
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    ba = BundleAdjuster(K, window_size=p.n_cams, ftol=1e-12, xtol=1e-12, pcg_tol=1e-3)
    buf = io.StringIO()
    with redirect_stdout(buf):
        ba.run(gmap)
    ba.close()
    assert "LBA Complete" in buf.getvalue()
    ids = sorted(gmap.map_points)
    pts = np.array([gmap.map_points[i].position.ravel() for i in ids])
    Rs = np.array([gmap.keyframes[k].R for k in sorted(gmap.keyframes)])
    got = [(np.load(tmp_path / f"run_pts_{k}.npy"), np.load(tmp_path / f"run_R_{k}.npy"),
            open(tmp_path / f"run_log_{k}.txt").read()) for k in range(2)]
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]) and got[0][2] == got[1][2]
    assert got[0][2] == buf.getvalue()                       # same log line: costs agree to the printed cents
    assert np.abs(got[0][0] - pts).max() <= 1e-5 and np.abs(got[0][1] - Rs).max() <= 1e-7


SHARD_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.problem import extract_shard, shard_by_landmark
from bundle_adjustment_amd.synthetic import make_config, make_problem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
mode = %(mode)r
if mode == "empty":
    p = make_problem(9, 700, 4, seed=23, outlier_frac=0.01)
    b, e = (0, p.n_pts) if rank == 0 else (p.n_pts, p.n_pts)          # rank 1 owns no landmark at all
    kw = dict(loss="huber", max_iters=12, ftol=1e-13, xtol=1e-13, gtol=1e-12, pcg_tol=1e-3)
else:
    p = make_config("C3", seed=0)
    b, e = shard_by_landmark(p, world)[rank]
    kw = dict(loss="huber", max_iters=4, ftol=0.0, xtol=0.0, gtol=1e-300, pcg_tol=1e-10, pcg_max_iters=300, pcg_model_tol=0.0)
sub, _ = extract_shard(p, b, e)
s = hip_backend.Solver(0)
uid = [hip_backend.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
s.comm_init(rank, world, uid[0])
s.set_problem(sub)
out = s.solve(**kw)
cams, pts = s.get_params()
np.save(os.path.join(%(out)r, f"cams_{rank}.npy"), cams)
np.save(os.path.join(%(out)r, f"pts_{rank}.npy"), pts)
json.dump(out, open(os.path.join(%(out)r, f"out_{rank}.json"), "w"))
s.close()
dist.barrier()
dist.destroy_process_group()
"""


def _run_two_ranks(tmp_path, mode, ipc="0"):
    script = tmp_path / f"shard_worker_{mode}.py"
    script.write_text(SHARD_WORKER % dict(root=ROOT, out=str(tmp_path), mode=mode))
    env = dict(os.environ, BA_COMM="shm", BA_IPC=ipc)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.load(open(tmp_path / f"out_{k}.json")) for k in range(2)]


@pytest.mark.parametrize("ipc", ["0", "1"])
def test_rank_with_an_empty_landmark_shard_joins_every_collective(tmp_path, ipc):
    """One rank owns every landmark, the other none (what a skewed track distribution or world > n_pts can produce):
    the empty rank must still take part in every all-reduce of ba_solve -- no deadlock -- and both ranks must report
    the single-rank result."""
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.synthetic import make_problem
    outs = _run_two_ranks(tmp_path, "empty", ipc)
    p = make_problem(9, 700, 4, seed=23, outlier_frac=0.01)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        ref = s.solve(loss="huber", max_iters=12, ftol=1e-13, xtol=1e-13, gtol=1e-12, pcg_tol=1e-3)
        cams_ref, pts_ref = s.get_params()
    for key in ("iterations", "accepted", "pcg_iterations", "initial_sse", "final_sse", "final_cost", "status"):
        assert outs[0][key] == outs[1][key], key
    assert outs[0]["iterations"] == ref["iterations"]
    assert abs(outs[0]["final_cost"] - ref["final_cost"]) <= 1e-10 * ref["final_cost"]
    cams0, cams1 = np.load(tmp_path / "cams_0.npy"), np.load(tmp_path / "cams_1.npy")
    assert np.array_equal(cams0, cams1) and np.abs(cams0 - cams_ref).max() <= 1e-8
    assert np.load(tmp_path / "pts_1.npy").shape[0] == 0
    assert np.abs(np.load(tmp_path / "pts_0.npy") - pts_ref).max() <= 1e-7


@pytest.mark.parametrize("ipc", ["0", "1"])
def test_c3_sized_shard_pair_reproduces_the_single_rank_iterates(tmp_path, ipc):
    """SURVEY.md section 4, multi-GPU row: the same iterates to <= 1e-10 relative (all-reduce order aside), at the
    headline size: two landmark shards of C3 (500k observations each) against the single-rank run, four LM
    iterations with a tight PCG so that re-ordered sums are all that differs."""
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.problem import shard_by_landmark
    from bundle_adjustment_amd.synthetic import make_config
    outs = _run_two_ranks(tmp_path, "c3", ipc)                # (ipc = 1: the per-PCG-iteration exchange through IPC-mapped peer buffers)
    p = make_config("C3", seed=0)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        ref = s.solve(loss="huber", max_iters=4, ftol=0.0, xtol=0.0, gtol=1e-300, pcg_tol=1e-10, pcg_max_iters=300, pcg_model_tol=0.0)
        cams_ref, pts_ref = s.get_params()
    for key in ("iterations", "accepted", "initial_sse", "final_sse", "final_cost"):
        assert outs[0][key] == outs[1][key], key
    assert outs[0]["iterations"] == ref["iterations"] == 4 and outs[0]["accepted"] == ref["accepted"]
    assert abs(outs[0]["final_cost"] - ref["final_cost"]) <= 1e-10 * ref["final_cost"]
    cams0, cams1 = np.load(tmp_path / "cams_0.npy"), np.load(tmp_path / "cams_1.npy")
    assert np.array_equal(cams0, cams1)                     # replicated cameras stay bitwise identical across ranks
    assert np.abs(cams0 - cams_ref).max() <= 1e-10 * np.abs(cams_ref).max()
    ranges = shard_by_landmark(p, 2)
    pts = np.concatenate([np.load(tmp_path / f"pts_{k}.npy") for k in range(2)])
    assert ranges[1][1] == p.n_pts and pts.shape == pts_ref.shape
    assert np.abs(pts - pts_ref).max() <= 1e-10 * np.abs(pts_ref).max()


def test_bench_two_ranks_over_the_shm_transport_reports_its_transport(tmp_path):
    """bench.py --gpus 2 under torch.distributed.run, both ranks on the one GPU through BA_COMM=shm: the JSON line
    names the transport and how many ranks joined the communicator (what a SCALE record is checked against); without
    BA_COMM=shm two ranks on one device must make RCCL fail and the bench exit NON-ZERO instead of falling back."""
    env = dict(os.environ, BA_COMM="shm")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--config", "C2", "--repeats", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["world"] == 2 and cfg["transport"] == "shm" and cfg["ranks_in_communicator"] == 2
    assert line["steps"] == 4 and line["value"] > 0 and line["repeats"] == 2
    # next to the strong line: the problem that grows with the ranks (every rank its own C2-sized landmark shard)
    weak = line["weak_scaling"]
    assert weak["value"] > 0 and "x 2" in weak["workload"] and f"{2 * 5000} pts" in weak["workload"] and weak["final_rmse_px"] < 3.0


def test_bench_starts_its_own_ranks_when_run_without_a_launcher():
    """VERDICT r3, missing 1: `python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment -- the way
    the driver runs the one-GPU line -- must start its two ranks itself (children of a parent that makes no GPU call) and
    print a line for TWO ranks; it must never time one GPU and call it two.  Both ranks share the one GPU through
    BA_COMM=shm.  Over RCCL the same command finds one device for two ranks and must exit NON-ZERO with an error object
    instead of a result line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "C2",
           "--repeats", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=dict(env, BA_COMM="shm"), capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world"] == 2 and line["config"]["ranks_in_communicator"] == 2
    assert line["steps"] == 3 and line["value"] > 0
    env.pop("BA_COMM", None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert r.returncode != 0
    objs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert all("error" in ob for ob in objs) and not any("value" in ob for ob in objs)


def test_real_rccl_communicator_of_one_rank_runs_the_whole_multi_rank_loop(monkeypatch):
    """RCCL refuses two ranks on one device and only a one-GPU box is available, so until the driver's multi-GPU run the
    library's ncclAllReduce call site would never execute.  BA_COMM_FORCE=1 makes ba_comm_init build a REAL RCCL
    communicator of one rank and switches the solve onto the multi-rank control flow: every fold kernel, every
    ncclAllReduce (linearisation partials, Schur product per PCG iteration, gradient maximum, step scalars), the
    device-side decision kernel and ba_allgather_points run through librccl on the GPU.  With one rank an all-reduce is
    a copy and the folded partial sums add up in the same order, so the result must equal the plain single-rank solve
    bit for bit."""
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.synthetic import make_problem
    p = make_problem(12, 1500, 5, seed=41, outlier_frac=0.02)
    kw = dict(loss="huber", max_iters=15, ftol=1e-12, xtol=1e-12, gtol=1e-10, pcg_tol=1e-2)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        ref = s.solve(**kw)
        cams_ref, pts_ref = s.get_params()
    monkeypatch.delenv("BA_COMM", raising=False)
    monkeypatch.setenv("BA_COMM_FORCE", "1")
    uid = hip_backend.comm_unique_id()                        # ncclGetUniqueId
    with hip_backend.Solver(0) as s:
        s.comm_init(0, 1, uid)                                # ncclCommInitRank, world of one
        s.set_problem(p)
        r, sse, _ = s.residuals("huber")                      # first collective
        out = s.solve(**kw)
        cams, pts = s.get_params()
        allpts = s.allgather_points(0, p.n_pts)
        prof_out = s.solve(profile=1, **dict(kw, max_iters=2))
        prof = s.profile()
    for key in ("iterations", "accepted", "pcg_iterations", "initial_sse", "final_sse", "final_cost", "status"):
        assert out[key] == ref[key], key
    assert np.array_equal(cams, cams_ref) and np.array_equal(pts, pts_ref) and np.array_equal(allpts, pts_ref)
    assert prof["allreduce"]["launches"] > 0 and prof_out["iterations"] == 2          # the collectives really ran


BAL_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.problem import BAProblem, extract_shard, shard_by_landmark
from bundle_adjustment_amd.synthetic import make_bal_problem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
bal = make_bal_problem(60, 5000, 22000, seed=5)
p = BAProblem(np.ascontiguousarray(bal.cams[:, :6]), bal.pts, bal.cam_idx, bal.pt_idx, bal.uv, np.array([1.0, 1.0, 0.0, 0.0]), 0)
b, e = shard_by_landmark(p, world)[rank]
sub, _ = extract_shard(p, b, e)
s = hip_backend.Solver(0)
uid = [hip_backend.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
s.comm_init(rank, world, uid[0])
s.set_problem(sub)
intr = np.ascontiguousarray(bal.cams[:, 6:9]).copy()
out = s.solve_bal_resident(intr, loss="huber", max_iters=12, ftol=0.0, xtol=0.0, gtol=0.0, pcg_tol=1e-3, pcg_max_iters=400)
cams, pts = s.get_params()
np.save(os.path.join(%(out)r, f"cams_{rank}.npy"), np.concatenate([cams, intr], axis=1))
np.save(os.path.join(%(out)r, f"pts_{rank}.npy"), pts)
out["ipc_exchanges"] = s.stats()["ipc_exchanges"]
json.dump(out, open(os.path.join(%(out)r, f"out_{rank}.json"), "w"))
s.close()
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("ipc", ["0", "1"])
def test_two_ranks_bal_camera_match_single_rank(tmp_path, ipc):
    """The BAL 9-parameter camera shards by landmark like the pinhole (fold / all-reduce sizes follow the camera model: 9 and
    45 + 9 sums per camera): two ranks on one GPU (shm transport) against the single-rank ba_solve_bal -- same global costs on
    both ranks, cameras (f, k1, k2 included) bitwise identical across ranks, and the single-rank iterates to round-off.
    ipc = 1: the per-PCG-iteration exchange inside k_pcg_step (records of 2 + 9 x 16 doubles per workgroup here)."""
    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.synthetic import make_bal_problem
    script = tmp_path / "worker_bal.py"
    script.write_text(BAL_WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, BA_COMM="shm", BA_IPC=ipc)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    bal = make_bal_problem(60, 5000, 22000, seed=5)
    with hip_backend.Solver(0) as s:
        ref, cams_ref, pts_ref = s.solve_bal(bal, fixed_cam=0, loss="huber", max_iters=12, ftol=0.0, xtol=0.0, gtol=0.0, pcg_tol=1e-3,
                                             pcg_max_iters=400)
    outs = [json.load(open(tmp_path / f"out_{k}.json")) for k in range(2)]
    for key in ("iterations", "accepted", "pcg_iterations", "initial_sse", "final_sse", "final_cost"):
        assert outs[0][key] == outs[1][key], key
    assert (outs[0]["ipc_exchanges"] >= outs[0]["pcg_iterations"]) if ipc == "1" else (outs[0]["ipc_exchanges"] == 0)
    assert abs(outs[0]["initial_cost"] - ref["initial_cost"]) <= 1e-10 * ref["initial_cost"]
    assert abs(outs[0]["final_cost"] - ref["final_cost"]) <= 1e-8 * ref["final_cost"]
    assert outs[0]["final_cost"] < 0.05 * outs[0]["initial_cost"]
    c0, c1 = np.load(tmp_path / "cams_0.npy"), np.load(tmp_path / "cams_1.npy")
    assert np.array_equal(c0, c1) and c0.shape == (60, 9)
    assert np.abs(c0 - cams_ref).max() <= 1e-6 * np.abs(cams_ref).max()
    pts = np.concatenate([np.load(tmp_path / f"pts_{k}.npy") for k in range(2)])
    assert pts.shape == pts_ref.shape and np.abs(pts - pts_ref).max() <= 1e-5 * np.abs(pts_ref).max()
