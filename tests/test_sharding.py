"""Landmark sharding.  CPU: the shard bookkeeping, and -- with two gloo ranks
-- that all-reducing per-shard reduced camera systems reproduces the full system (oracle as
the checker).  GPU: shards computed through the C ABI sum to the whole."""
import os
import socket

import numpy as np
import pytest

from bundle_adjustment_amd.problem import extract_shard, shard_by_landmark
from bundle_adjustment_amd.synthetic import make_problem
from oracle import ba_oracle as o


def test_shards_partition_points_and_observations():
    p = make_problem(9, 500, 4, seed=1)
    for g in (1, 2, 3, 8):
        ranges = shard_by_landmark(p, g)
        assert ranges[0][0] == 0 and ranges[-1][1] == p.n_pts
        seen = np.zeros(p.n_obs, dtype=int)
        counts = []
        for (b, e), nxt in zip(ranges, ranges[1:] + [(p.n_pts, p.n_pts)]):
            assert e == nxt[0] and b <= e
            sub, sel = extract_shard(p, b, e)
            seen[sel] += 1
            counts.append(sub.n_obs)
            assert sub.n_cams == p.n_cams and sub.n_pts == e - b
            np.testing.assert_array_equal(sub.pts, p.pts[b:e])
            np.testing.assert_array_equal(sub.pt_idx + b, p.pt_idx[sel])
            np.testing.assert_array_equal(sub.uv, p.uv[sel])
        assert np.all(seen == 1)
        assert max(counts) - min(counts) <= 2 * 4 + 4          # balanced by observation count


def test_skewed_tracks_never_give_an_empty_shard():
    """A few very long tracks can pull two cuts onto the same point; with at least as many points as shards every
    shard must still own a point (an empty rank is legal for the library but wastes a GPU)."""
    from bundle_adjustment_amd.problem import BAProblem
    K4 = np.array([500.0, 500.0, 320.0, 240.0])
    for lens, world in (([1, 1, 10], 2), ([40, 1, 1, 1], 4), ([1, 1, 1, 90], 3), ([5] * 8, 8), ([1, 200], 2)):
        npt = len(lens)
        pt_idx = np.repeat(np.arange(npt, dtype=np.int32), lens)
        cam_idx = np.concatenate([np.arange(n, dtype=np.int32) % 3 for n in lens])
        p = BAProblem(np.zeros((3, 6)), np.zeros((npt, 3)), cam_idx, pt_idx, np.zeros((pt_idx.size, 2)), K4, 0)
        ranges = shard_by_landmark(p, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == npt
        assert all(e > b for b, e in ranges), (lens, world, ranges)
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    # fewer points than shards: the trailing shards are empty, the leading ones keep one point each
    p = BAProblem(np.zeros((3, 6)), np.zeros((2, 3)), np.array([0, 1], dtype=np.int32), np.array([0, 1], dtype=np.int32),
                  np.zeros((2, 2)), K4, 0)
    ranges = shard_by_landmark(p, 4)
    assert ranges[0][0] == 0 and ranges[-1][1] == 2 and sum(e - b for b, e in ranges) == 2


def test_run_refuses_fewer_landmarks_than_ranks():
    """SPMD run(): a window with fewer landmarks than ranks raises the same error on every rank before any collective."""
    from bundle_adjustment_amd import BundleAdjuster
    from bundle_adjustment_amd.synthetic import problem_to_map
    from tests.fake_solver import GlooOracleSolver
    p = make_problem(3, 2, 2, seed=0)
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    ba = BundleAdjuster(K, window_size=p.n_cams, comm=(0, 4, b"x" * 128))
    ba._solver = GlooOracleSolver()
    ba._solver.comm_init(0, 4, b"")
    with pytest.raises(ValueError, match="cannot be sharded"):
        ba.run(problem_to_map(p))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    p = make_problem(7, 300, 4, seed=5)
    b, e = shard_by_landmark(p, world)[rank]
    sub, _ = extract_shard(p, b, e)
    lam = 1e-3
    ne = o.normal_equations(sub.cams, sub.pts, sub.cam_idx, sub.pt_idx, sub.uv, sub.K4, 0, "huber")
    # what every rank all-reduces once per linearisation: Hcc | bc
    hb = torch.from_numpy(np.concatenate([ne["Hcc"].ravel(), ne["bc"].ravel()]))
    dist.all_reduce(hb)
    nc = p.n_cams
    ne["Hcc"] = hb[:36 * nc].numpy().reshape(nc, 6, 6).copy()
    ne["bc"] = hb[36 * nc:].numpy().reshape(nc, 6).copy()
    # per PCG iteration: the shard's W Hpp^-1 W^T v (and the u.y word) -- all-reduced; S v assembled from it
    op = o.SchurOperator(ne, sub.cam_idx, sub.pt_idx, lam, 0)
    v = np.random.default_rng(0).normal(size=(nc, 6))
    v[0] = 0
    wy = torch.from_numpy(op.w_times(op.wt_times(v)))
    dist.all_reduce(wy)
    sv = np.einsum("cij,cj->ci", op.Hccd, v) - wy.numpy()
    sv[0] = v[0]
    # right-hand side
    y0 = np.einsum("pij,pj->pi", op.Hppinv, ne["bp"])
    wy0 = torch.from_numpy(op.w_times(y0))
    dist.all_reduce(wy0)
    g = -(ne["bc"] - wy0.numpy())
    g[0] = 0
    if rank == 0:
        full = o.normal_equations(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber")
        fop = o.SchurOperator(full, p.cam_idx, p.pt_idx, lam, 0)
        q.put((float(np.abs(sv - fop.apply(v)).max() / np.abs(fop.apply(v)).max()),
               float(np.abs(g - fop.rhs()).max() / np.abs(fop.rhs()).max()),
               float(np.abs(ne["Hcc"] - full["Hcc"]).max() / np.abs(full["Hcc"]).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_of_reduced_camera_system():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    errs = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert max(errs) <= 1e-12, errs


@pytest.mark.gpu
def test_gpu_shards_sum_to_the_whole():
    """Through the C ABI: Hcc/bc of the shards add up to the full problem's, and at lambda = 0
    so do the Schur products (each shard contributes its own W Hpp^-1 W^T)."""
    from bundle_adjustment_amd import hip_backend
    p = make_problem(10, 600, 4, seed=8)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        Hcc, bc, _, _ = s.linearize("huber")
        v = np.random.default_rng(1).normal(size=(p.n_cams, 6))
        v[0] = 0
        sv = s.schur_apply(0.0, v)
        acc_H, acc_b, acc_sv = np.zeros_like(Hcc), np.zeros_like(bc), np.zeros_like(sv)
        for b, e in shard_by_landmark(p, 3):
            sub, _ = extract_shard(p, b, e)
            s.set_problem(sub)
            h, bb, _, _ = s.linearize("huber")
            acc_H += h
            acc_b += bb
            acc_sv += s.schur_apply(0.0, v)
        assert np.abs(acc_H - Hcc).max() <= 1e-10 * np.abs(Hcc).max()
        assert np.abs(acc_b - bc).max() <= 1e-10 * np.abs(bc).max()
        acc_sv[0] = sv[0]
        assert np.abs(acc_sv - sv).max() <= 1e-9 * np.abs(sv).max()


def _spmd_run_main(rank, world, port, q):
    """One rank of an SPMD BundleAdjuster.run: identical maps in, identical maps out."""
    import io
    from contextlib import redirect_stdout
    import torch.distributed as dist
    from bundle_adjustment_amd import bundle_adjuster as ba_mod
    from bundle_adjustment_amd.synthetic import problem_to_map
    from tests.fake_solver import GlooOracleSolver
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ba_mod.hip_backend.Solver = GlooOracleSolver
    p = make_problem(6, 240, 4, seed=13)
    gmap = problem_to_map(p)
    K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
    ba = ba_mod.BundleAdjuster(K, window_size=p.n_cams, comm=(rank, world, b"\0" * 128))
    buf = io.StringIO()
    with redirect_stdout(buf):
        ba.run(gmap)
    ids = sorted(gmap.map_points)
    pts = np.array([gmap.map_points[i].position.ravel() for i in ids])
    Rs = np.array([gmap.keyframes[k].R for k in sorted(gmap.keyframes)])
    q.put((rank, buf.getvalue(), pts, Rs, ba.last_summary["final_sse"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_spmd_run_matches_single_rank_host_logic():
    """CPU, gloo, world 2: BundleAdjuster.run with comm=(rank, world, id) shards by landmark, gathers
    every shard's points and leaves the same map on both ranks as a single-rank run (device replaced by
    the oracle-backed doubles of tests/fake_solver.py)."""
    import io
    from contextlib import redirect_stdout
    import torch.multiprocessing as mp
    from bundle_adjustment_amd import bundle_adjuster as ba_mod
    from bundle_adjustment_amd.synthetic import problem_to_map
    from tests.fake_solver import OracleSolver
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_spmd_run_main, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # single rank, same double
    saved = ba_mod.hip_backend.Solver
    ba_mod.hip_backend.Solver = OracleSolver
    try:
        p = make_problem(6, 240, 4, seed=13)
        gmap = problem_to_map(p)
        K = np.array([[p.K4[0], 0, p.K4[2]], [0, p.K4[1], p.K4[3]], [0, 0, 1.0]])
        ba = ba_mod.BundleAdjuster(K, window_size=p.n_cams)
        buf = io.StringIO()
        with redirect_stdout(buf):
            ba.run(gmap)
    finally:
        ba_mod.hip_backend.Solver = saved
    ids = sorted(gmap.map_points)
    pts = np.array([gmap.map_points[i].position.ravel() for i in ids])
    Rs = np.array([gmap.keyframes[k].R for k in sorted(gmap.keyframes)])
    assert "LBA Complete" in buf.getvalue()
    for rank, log, rpts, rRs, sse in got:
        assert log == buf.getvalue()
        np.testing.assert_allclose(rpts, pts, rtol=0, atol=1e-9)     # the shards re-order the sums
        np.testing.assert_allclose(rRs, Rs, rtol=0, atol=1e-9)
        assert abs(sse - ba.last_summary["final_sse"]) <= 1e-9 * sse
    np.testing.assert_array_equal(got[0][2], got[1][2])              # both ranks end with the same map
    np.testing.assert_array_equal(got[0][3], got[1][3])


def test_bench_without_a_launcher_refuses_instead_of_timing_one_rank():
    """CPU (no GPU in this container): `python bench.py --gpus 2` with no WORLD_SIZE starts its two ranks itself; they find
    no device, so the command must end NON-ZERO and print no result line -- in particular not a line for one GPU, which
    is what round 3's bench silently produced when it was run without a launcher.  A launcher environment that disagrees
    with --gpus is refused as well."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--config", "C1",
           "--repeats", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            assert "value" not in json.loads(ln)
    r = subprocess.run(cmd, env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120, cwd=root)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
