#!/usr/bin/env python3
"""Headline benchmark: LM iterations/sec + final reprojection RMSE on the synthetic
1k-camera / 100k-point / 1M-observation problem (BASELINE.json ``configs[2]``).

    python bench.py --gpus N --steps K --warmup W

A "step" is one Levenberg-Marquardt iteration (linearise, damped Schur system, PCG
solve, back substitution, trial-point cost, accept/reject) of the HIP solver.  The timed
region is one ``ba_solve`` call with ``max_iters = K`` and every stopping test disabled
(ftol = xtol = 0, gtol = 1e-300: positive, so the per-iteration gradient-norm kernel of a
production run is launched and timed), so exactly K iterations run; inputs (observation
lists, parameters) are resident in HBM before it starts and the call returns after the
stream has drained.  The K-step solve is repeated ``--repeats`` times (default 11) from the
same initial guess, each repeat bracketed by barrier + device synchronise; ``value`` is the
MEDIAN repeat (``value_min`` / ``value_max`` = slowest / fastest repeat).  For N > 1 the
driver launches one rank per GPU with torch.distributed.run; the points (and their
observations) are sharded by landmark block, cameras are replicated and the reduced
camera system is all-reduced with RCCL inside the library (torch.distributed is used
only to ship the 128-byte RCCL id, for the barriers and for the max over ranks).

One JSON line is printed by rank 0 (fields: see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(n_cams, n_pts, n_obs, cam_params=6):
    """SURVEY.md 8(d) minimum-traffic model, fp64, int32 indices (bytes).  cam_params: 6 for the reference's pinhole
    (rvec | t), 9 for the BAL camera (rvec | t | f k1 k2): P = 24 Np + 8 cam_params Nc, V = 8 cam_params Nc, and the
    packed camera blocks grow from 21 to 45 doubles."""
    P = 24 * n_pts + 8 * cam_params * n_cams
    V = 8 * cam_params * n_cams
    Q = 24 * n_pts
    D = 48 * n_pts
    hcc = 8 * (cam_params * (cam_params + 1) // 2) * n_cams
    b = dict(
        schur_pt=8 * n_obs + P + D + Q + V,            # index pass by point: W^T v, Hpp^-1
        schur_cam=8 * n_obs + P + Q + 3 * V,           # index pass by camera: W y, S v assembly
        lin=24 * n_obs + P + D + Q + hcc + V,
        back=8 * n_obs + P + D + 2 * Q + V,
        evalc=24 * n_obs + P,
    )
    b["pcg_iter"] = b["schur_pt"] + b["schur_cam"]
    return b


def cpu_baseline_port(prob, budget_s=15.0):
    """Reference CPU path (per-observation Python loop + 2-point finite differences over
    colour groups + TRF/LSMR, src/bundle_adjuster.py:24-72,170-174) timed on this host,
    single thread, on a bounded sample: the residual sweep is timed on a contiguous
    sample of observations and scaled to the full list; one TRF iteration costs
    (n_groups + 1) sweeps (scipy/optimize/_numdiff.py:628-705) plus an LSMR solve that is
    NOT timed (so the CPU figure is an upper bound on its throughput)."""
    from oracle import ba_oracle as o
    nobs = prob.n_obs
    n_sample = min(nobs, 1_000_000)           # C3: one full residual sweep, about 10 s of CPU work
    obs = [(int(c), int(p)) for c, p in zip(prob.cam_idx[:n_sample], prob.pt_idx[:n_sample])]
    kp = {ob: (float(u), float(v)) for ob, (u, v) in zip(obs, prob.uv[:n_sample])}
    x0, adj = o.pack_reference_params(prob.cams, prob.pts, prob.fixed_cam)
    K = np.array([[prob.K4[0], 0, prob.K4[2]], [0, prob.K4[1], prob.K4[3]], [0, 0, 1.0]])
    fixed_pose = (o.rodrigues_to_mat(prob.cams[prob.fixed_cam, :3]), prob.cams[prob.fixed_cam, 3:].reshape(3, 1))
    t0 = time.perf_counter()
    done = 0
    chunk = 20_000
    while done < n_sample and (time.perf_counter() - t0) < budget_s:
        sl = obs[done:done + chunk]
        o.reference_cost_function(x0, fixed_pose, prob.fixed_cam, adj, list(range(prob.n_pts)), sl, kp, K)
        done += len(sl)
    dt = time.perf_counter() - t0
    per_obs = dt / max(done, 1)
    from scipy.optimize._numdiff import group_columns
    tg = time.perf_counter()
    A = o.flat_sparsity(prob.n_cams, prob.n_pts, prob.cam_idx, prob.pt_idx, prob.fixed_cam)
    n_groups = int(group_columns(A).max()) + 1        # what least_squares does with jac_sparsity
    tg = time.perf_counter() - tg
    t_iter = (n_groups + 1) * per_obs * nobs
    # LSMR (scipy/sparse/linalg/_isolve/lsmr.py, what tr_solver='lsmr' runs): a few iterations on the
    # analytic CSR Jacobian at x0, for scale only -- the solve is NOT part of `value`
    from scipy.sparse.linalg import lsmr
    tl = time.perf_counter()
    J = o.flat_jacobian_fun(prob.cams, prob.n_pts, prob.cam_idx, prob.pt_idx, prob.K4, prob.fixed_cam)(x0)
    f0 = o.residuals(prob.cams, prob.pts, prob.cam_idx, prob.pt_idx, prob.uv, prob.K4).ravel()
    t_build = time.perf_counter() - tl
    tl = time.perf_counter()
    n_lsmr = 10
    lsmr(J, f0, maxiter=n_lsmr, atol=1e-6, btol=1e-6)
    t_lsmr = (time.perf_counter() - tl) / n_lsmr
    return dict(value=1.0 / t_iter, unit="LM iterations/s", cores=1, kind="port", n_groups=n_groups,
                sample=f"{done} of {nobs} observations through the per-observation loop "
                       f"({per_obs * 1e6:.1f} us/obs, {dt:.1f} s), x ({n_groups}+1) sweeps per TRF iteration; "
                       f"colour groups counted by scipy group_columns in {tg:.1f} s; LSMR solve not included "
                       f"(for scale: {t_lsmr:.2f} s per LSMR iteration on the {J.shape[0]}x{J.shape[1]} Jacobian, "
                       f"{n_lsmr} timed, CSR built in {t_build:.1f} s); host has {os.cpu_count()} cores")


def cpu_baseline_vectorised(prob, n_groups, budget_s=10.0):
    """BASELINE.md section 3, second line: the same scipy algorithm with a VECTORISED numpy residual
    (one sweep over all observations) instead of the per-observation loop; again (n_groups + 1)
    sweeps per TRF iteration, LSMR not timed.  Reported beside the faithful baseline for honesty."""
    from oracle import ba_oracle as o
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 1.0 and n < 20):
        o.residuals(prob.cams, prob.pts, prob.cam_idx, prob.pt_idx, prob.uv, prob.K4)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    t_sweep = (time.perf_counter() - t0) / n
    return dict(value=1.0 / ((n_groups + 1) * t_sweep), unit="LM iterations/s", cores=1,
                kind="port", sample=f"{n} vectorised sweeps over all {prob.n_obs} observations "
                                    f"({t_sweep * 1e3:.1f} ms each), x ({n_groups}+1) sweeps per TRF iteration; LSMR not timed")


def cpu_baseline_bal(bal, budget_s=15.0):
    """--camera bal: the reference has no BAL camera (its only camera is cv2.projectPoints(..., distCoeffs=None),
    src/bundle_adjuster.py:67), so the CPU figure is the reference's ALGORITHM (scipy TRF with 2-point finite differences over
    the colour groups of the 0/1 pattern: (n_groups + 1) residual sweeps per iteration, LSMR not timed) on the oracle's
    vectorised BAL residual -- one host core, numpy."""
    from oracle import ba_oracle as o
    from scipy.optimize._numdiff import group_columns
    from scipy.sparse import coo_matrix
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 5.0 and n < 50):
        o.bal_residuals(bal.cams, bal.pts, bal.cam_idx, bal.pt_idx, bal.uv)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    t_sweep = (time.perf_counter() - t0) / n
    nobs, nc = bal.n_obs, bal.n_cams
    rows, cols = [], []
    r2 = np.repeat(2 * np.arange(nobs), 2) + np.tile([0, 1], nobs)
    free = np.repeat(bal.cam_idx != 0, 2)
    for j in range(9):
        rows.append(r2[free]); cols.append(np.repeat(9 * bal.cam_idx.astype(np.int64) + j, 2)[free])
    for j in range(3):
        rows.append(r2); cols.append(np.repeat(9 * nc + 3 * bal.pt_idx.astype(np.int64) + j, 2))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    tg = time.perf_counter()
    A = coo_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(2 * nobs, 9 * nc + 3 * bal.n_pts)).tocsr()
    n_groups = int(group_columns(A).max()) + 1
    tg = time.perf_counter() - tg
    return dict(value=1.0 / ((n_groups + 1) * t_sweep), unit="LM iterations/s", cores=1, kind="port", n_groups=n_groups,
                sample=f"{n} vectorised numpy sweeps of the BAL residual over all {nobs} observations ({t_sweep * 1e3:.1f} ms each), "
                       f"x ({n_groups}+1) sweeps per TRF iteration (colour groups of the 2x9 / 2x3 pattern by scipy group_columns, "
                       f"{tg:.1f} s); LSMR not timed; host has {os.cpu_count()} cores")


class _stdout_to_stderr:
    """RCCL prints a version banner on file descriptor 1 when a communicator comes up; stdout must carry exactly one JSON
    line, so the descriptor points at stderr while the transport is being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def launch_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as children -- the same
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    the driver uses -- from a parent that makes no GPU call at all.  Rank 0's stdout (the one JSON line, or the error
    object of a run that could not bring its transport up) is relayed; the return value is the launcher's exit code, or 4
    when the children ended 'successfully' without a JSON line that names N ranks."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL / device-memory sharing across processes needs here
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                                  # (stderr goes straight through)
        out = out.strip()
        if out.startswith("{"):
            line = out
            print(out, flush=True)
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        return rc
    try:
        ok = line is not None and json.loads(line).get("n_gpus") == n and "error" not in json.loads(line)
    except ValueError:
        ok = False
    if not ok:
        print(f"bench.py: the {n} ranks ended without a result line for n_gpus={n}", file=sys.stderr)
        return 4
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C5", "C3x10"])
    ap.add_argument("--camera", default="pinhole", choices=["pinhole", "bal"],
                    help="bal (with --config C5): BASELINE config 5 as stated -- the BAL 9-parameter camera [rvec | t | f k1 k2], "
                         "f / k1 / k2 distinct per camera and adjusted with the poses (ba_solve_bal)")
    ap.add_argument("--loss", default="huber", choices=["huber", "linear"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pcg-tol", type=float, default=0.1)
    ap.add_argument("--pcg-max-iters", type=int, default=200)
    ap.add_argument("--pcg-model-tol", type=float, default=-1.0,
                    help="Nash & Sofer model test of the PCG loop (ba_options.pcg_model_tol; 0 = off; -1 = the library's default: "
                         "0.5 on band-structured problems on one rank, else off)")
    ap.add_argument("--precond", default="schur_jacobi", choices=["schur_jacobi", "jacobi", "two_level"])
    ap.add_argument("--precond-lag", type=int, default=None,
                    help="ba_options.precond_lag: damped systems that may keep the Schur-Jacobi blocks of an earlier one (default: the library's)")
    ap.add_argument("--jacobian", default="f64", choices=["f64", "f32"], help="f32: config 5's fp32 Jacobian blocks in the PCG passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=11, help="timed repeats of the K-step solve; value = the median repeat")
    ap.add_argument("--weak", action="store_true",
                    help="also time the problem that GROWS with the ranks (every rank a C3-sized landmark shard) and report it "
                         "under 'weak_scaling'; on by default for --gpus > 1 with the headline config")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: this process starts the N ranks itself (before it has touched the GPU in any way),
        # relays rank 0's JSON line and exits with the launcher's code -- it never times ONE GPU under the name of N
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to time {world} rank(s) under the name of {args.gpus}")
    dist = None
    if world > 1:
        import torch.distributed as dist          # plumbing only: id exchange, barrier, max over ranks
        dist.init_process_group(backend="gloo")

    from bundle_adjustment_amd import hip_backend
    from bundle_adjustment_amd.problem import extract_shard, shard_by_landmark
    from bundle_adjustment_amd.synthetic import make_bal_like, make_config

    intr0 = None
    if args.camera == "bal":
        if args.config != "C5":
            raise SystemExit("--camera bal is BASELINE config 5: use it with --config C5")
        from bundle_adjustment_amd.problem import BAProblem
        from bundle_adjustment_amd.synthetic import make_bal_problem
        bal = make_bal_problem(seed=args.seed)
        # the handle's problem upload is camera-model agnostic (poses, points, observation lists); the per-camera
        # (f, k1, k2) travel with every ba_solve_bal call
        prob = BAProblem(np.ascontiguousarray(bal.cams[:, :6]), bal.pts, bal.cam_idx, bal.pt_idx, bal.uv, np.array([1.0, 1.0, 0.0, 0.0]), 0)
        intr0 = np.ascontiguousarray(bal.cams[:, 6:9])
    else:
        prob = make_bal_like(seed=args.seed) if args.config == "C5" else make_config(args.config, seed=args.seed)
    n_obs_total = prob.n_obs
    shard = prob
    if world > 1:
        b, e = shard_by_landmark(prob, world)[rank]
        shard, _ = extract_shard(prob, b, e)

    ndev = hip_backend.device_count()
    dev = (local_rank % ndev) if world > 1 else 0
    solver = hip_backend.Solver(dev)
    comm_note, transport, ranks_up = "none (single rank)", "none", 1
    if world > 1:
        import torch
        # Transport: RCCL over xGMI unless BA_COMM=shm was asked for (the host-staged test transport that lets several
        # ranks share one GPU).  There is NO fallback: if RCCL cannot be brought up on every rank the run exits non-zero
        # -- a scaling line over the wrong transport would be worse than none.
        transport = "shm" if os.environ.get("BA_COMM") == "shm" else "rccl"
        ok, err, uid = 1, "", None
        if transport == "rccl" and ndev < world:
            # one rank per GPU: RCCL refuses two ranks on one device, and a line over fewer GPUs than it names is worse than none
            if rank == 0:
                print(json.dumps({"error": f"--gpus {world} but only {ndev} device(s) visible (BA_COMM=shm lets ranks share a GPU for tests)",
                                  "n_gpus": world}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
        if rank == 0:                         # rank 0 ALWAYS reaches the broadcast, with or without an id
            try:
                with _stdout_to_stderr():
                    uid = hip_backend.comm_unique_id()
            except hip_backend.BAHipError as e:
                ok, err = 0, str(e)
        box = [(ok, uid)]
        dist.broadcast_object_list(box, src=0)
        ok0, uid = box[0]
        if ok0:
            try:
                with _stdout_to_stderr():
                    solver.comm_init(rank, world, uid)
                    solver.set_problem(shard)
                    solver.residuals("linear", want_vector=False)        # first collective: proves the transport works
            except hip_backend.BAHipError as e:
                ok, err = 0, str(e)
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)                  # how many ranks joined the communicator
        ranks_up = int(flag.item())
        if ranks_up != world:
            if rank == 0:
                print(json.dumps({"error": f"{transport} communicator came up on {ranks_up} of {world} ranks", "detail": err,
                                  "n_gpus": world}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
        comm_note = f"{transport}: communicator of {world} ranks initialised on {ranks_up} ranks, first all-reduce done"
    else:
        if os.environ.get("BA_COMM_FORCE"):
            # one GPU, but through the multi-rank control flow and a real RCCL communicator of one rank: what the fold
            # kernels, the RCCL launches and the device-side decision cost per LM iteration (not a scaling number)
            with _stdout_to_stderr():
                solver.comm_init(0, 1, hip_backend.comm_unique_id())
                solver.set_problem(shard)
                solver.residuals("linear", want_vector=False)
            transport = "shm" if os.environ.get("BA_COMM") == "shm" else "rccl"
            comm_note = f"{transport}: communicator of ONE rank forced (BA_COMM_FORCE), multi-rank control flow on one GPU"
        solver.set_problem(shard)

    # every stopping tolerance off so that exactly K iterations run; gtol is tiny but POSITIVE so that the
    # per-iteration gradient-norm work of a production run() (per-workgroup maxima folded by the first PCG probe) is inside
    # the timed region
    kw = dict(loss=args.loss, ftol=0.0, xtol=0.0, gtol=1e-300, pcg_tol=args.pcg_tol, pcg_max_iters=args.pcg_max_iters,
              pcg_model_tol=args.pcg_model_tol,
              preconditioner=args.precond, jacobian_precision=1 if args.jacobian == "f32" else 0)
    if args.precond_lag is not None:
        kw["precond_lag"] = args.precond_lag

    # the PCG model test in effect (ba_options.pcg_model_tol = -1 resolves inside the library; mirrored here for the line)
    # (the automatic default needs a loose outer tolerance, ftol >= 1e-6; the bench runs with ftol = 0: off)
    model_tol_eff = args.pcg_model_tol if args.pcg_model_tol >= 0 else 0.0

    def run_solve(**k):
        if intr0 is not None:
            return solver.solve_bal_resident(intr0.copy(), **k)      # (the copy: ba_solve_bal adjusts the intrinsics in place)
        return solver.solve(**k)

    def barrier():
        solver.synchronize()
        if dist is not None:
            dist.barrier()

    # warmup: W untimed LM iterations, then restore the initial guess
    if args.warmup > 0:
        run_solve(max_iters=args.warmup, **kw)
    # timed region: EXACTLY K LM iterations from the same initial guess, bracketed by barrier + device sync on both
    # sides; repeated `--repeats` times (the 5 ms of one repeat is one sample; value = the MEDIAN repeat, max over
    # ranks per repeat), parameters restored outside the timed region between repeats
    dts = []
    for _ in range(max(1, args.repeats)):
        solver.set_params(shard.cams, shard.pts)
        barrier()
        t0 = time.perf_counter()
        out = run_solve(max_iters=args.steps, **kw)
        solver.synchronize()
        dt_rep = time.perf_counter() - t0
        barrier()
        if dist is not None:
            import torch
            t = torch.tensor([dt_rep], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_rep = float(t.item())
        dts.append(dt_rep)
    dt = float(np.median(dts))
    steps_done = out["iterations"]
    rmse = float(np.sqrt(out["final_sse"] / n_obs_total))

    # per-kernel durations with HIP events on the solver's stream: same workload again
    solver.set_params(shard.cams, shard.pts)
    solver.profile(reset=True)
    run_solve(max_iters=args.steps, profile=1, **kw)
    prof = solver.profile()
    ab = algorithmic_bytes(prob.n_cams, shard.n_pts, shard.n_obs, 9 if intr0 is not None else 6)
    # The dominant kernel is chosen the way a rocprofv3 --stats summary of this command shows it: the PCG point pass's kernel
    # symbol also covers the launches that found PCG finished and went on as the back substitution ("schur_pt_then_backsub":
    # one per LM iteration); its roofline figure is taken over the pure PCG launches (17.7 MB each at C3), the fused ones
    # are reported next to it.
    fused = prof.get("schur_pt_then_backsub", {})

    def _total(k):
        return prof.get(k, {}).get("total_ms", 0.0) + (fused.get("total_ms", 0.0) if k == "schur_pt" else 0.0)
    dom = max(("schur_pt", "schur_cam"), key=_total)
    traffic, traffic_source = None, None
    if dom in prof and prof[dom].get("working_launches", 0) > 0:
        dom_us = prof[dom]["working_mean_us"]        # launches that exit at once after PCG convergence are left out
        achieved = ab[dom] / (dom_us * 1e-6) / 1e9
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and args.config == "C3" and os.path.exists(tpath):     # the PMC passes were run on this workload only
            try:
                tj = json.load(open(tpath))
                traffic, traffic_source = tj.get(dom), tj.get("_source", "profiles/traffic.json (rocprofv3 PMC passes of an earlier run of this command)")
            except Exception:
                traffic = None
        roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                        # traffic is NOT measured by this run: PMC counters need their own rocprofv3 passes (the guide's rule)
                        traffic_source=traffic_source,
                        algorithmic_bytes_per_launch=ab[dom], mean_launch_us=round(dom_us, 3),
                        launches=prof[dom]["working_launches"], early_exit_launches=prof[dom]["launches"] - prof[dom]["working_launches"],
                        # what rocprofv3 --stats averages: every launch of the kernel, early exits included
                        mean_launch_us_all_launches=round(prof[dom]["mean_us"], 3))
        if dom == "schur_pt" and fused.get("launches", 0) > 0:
            roofline["launches_that_went_on_as_back_substitution"] = dict(
                launches=fused["launches"], mean_launch_us=round(fused["mean_us"], 3), algorithmic_bytes_per_launch=ab["back"])
    else:
        # a window-sized problem (<= 8 cameras, <= 6144 observations): ba_solve ran it as ONE launch of k_small_lm
        # (csrc/ba_small.hpp), there is no PCG pass to price; the kernel is latency-bound inside one compute unit, its
        # "roofline" is the whole solve's algorithmic bytes over its duration
        k_us = prof.get("misc", {}).get("mean_us", 0.0)
        b_solve = (ab["lin"] + ab["back"] + ab["evalc"]) * max(out["iterations"], 1)
        achieved = b_solve / (k_us * 1e-6) / 1e9 if k_us > 0 else 0.0
        roofline = dict(bound="hbm", kernel="window solver (whole solve in one launch: k_small_mw / k_small_lm)", achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 6), traffic=None, algorithmic_bytes_per_launch=round(b_solve),
                        mean_launch_us=round(k_us, 3), launches=prof.get("misc", {}).get("launches", 0), early_exit_launches=0,
                        mean_launch_us_all_launches=round(k_us, 3))
    # SURVEY.md 8(d)'s whole-iteration figure: B_iter = B_lin + K_pcg B_pcg + B_back + B_eval over the timed wall time
    k_pcg = out["pcg_iterations"] / max(steps_done, 1)
    b_iter = ab["lin"] + k_pcg * ab["pcg_iter"] + ab["back"] + ab["evalc"]
    iter_gbs = b_iter / (dt / max(steps_done, 1)) / 1e9
    roofline["lm_iteration"] = dict(algorithmic_bytes=round(b_iter), achieved=round(iter_gbs, 1),
                                    frac=round(iter_gbs / HBM_PEAK_GBS, 4))

    # ---- weak scaling beside the strong headline (one exchange per PCG iteration makes the FIXED 1k / 100k
    # problem latency-bound across GPUs; the same exchange amortises over N times the landmarks): every rank generates its
    # own C3-sized landmark shard of an N x 100k-point problem (cameras identical on all ranks) and the same K steps are timed
    weak = None
    if (args.weak or world > 1) and args.camera == "pinhole" and args.config in ("C2", "C3"):
        from bundle_adjustment_amd.synthetic import CONFIGS, make_weak_shard
        cfgw = CONFIGS[args.config]
        wshard = make_weak_shard(cfgw["n_cams"], cfgw["n_pts"], cfgw["obs_per_pt"], args.seed, rank)
        solver.set_problem(wshard)
        if args.warmup > 0:
            solver.solve(max_iters=args.warmup, **kw)
        wdts = []
        for _ in range(max(1, min(args.repeats, 5))):
            solver.set_params(wshard.cams, wshard.pts)
            barrier()
            t0 = time.perf_counter()
            wout = solver.solve(max_iters=args.steps, **kw)
            solver.synchronize()
            dtw = time.perf_counter() - t0
            barrier()
            if dist is not None:
                import torch
                t = torch.tensor([dtw], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtw = float(t.item())
            wdts.append(dtw)
        n_obs_w = wshard.n_obs
        if dist is not None:
            import torch
            t = torch.tensor([float(n_obs_w)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            n_obs_w = int(t.item())
        dtw = float(np.median(wdts))
        weak = dict(value=round(wout["iterations"] / dtw, 3), unit="LM iterations/s", ms_per_step=round(1e3 * dtw / max(wout["iterations"], 1), 4),
                    workload=f"{args.config} x {world}: {cfgw['n_cams']} cams / {cfgw['n_pts'] * world} pts / {n_obs_w} obs "
                             f"({cfgw['n_pts']} landmarks per rank, generated per rank)",
                    pcg_iterations_per_lm=round(wout["pcg_iterations"] / max(wout["iterations"], 1), 2),
                    final_rmse_px=round(float(np.sqrt(wout["final_sse"] / n_obs_w)), 6), repeats=len(wdts))

    line = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_bal(bal) if intr0 is not None else cpu_baseline_port(prob)
        line = {
            "metric": "LM iterations/sec + final reprojection RMSE, 1k cams / 100k pts",   # BASELINE.json; RMSE: config.final_rmse_px
            "value": round(steps_done / dt, 3), "unit": "LM iterations/s", "n_gpus": world, "steps": steps_done,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / max(steps_done, 1), 4),
            "repeats": len(dts), "value_min": round(steps_done / max(dts), 3), "value_max": round(steps_done / min(dts), 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.jacobian == "f64" else "f64 accumulation and solve, f32 Jacobian blocks in the PCG passes",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {prob.n_cams} cams / {prob.n_pts} pts / {n_obs_total} obs, "
                                   f"loss={args.loss}, LM+Schur+PCG(tol {args.pcg_tol}, model test {model_tol_eff})"
                                   + (", BAL 9-parameter camera (f, k1, k2 per camera, adjusted)" if intr0 is not None else ""),
                       "camera": args.camera,
                       "parallelism": f"landmark-sharded x{world}" if world > 1 else "single GPU", "comm": comm_note,
                       "world": world, "transport": transport, "ranks_in_communicator": ranks_up,
                       "devices_visible": ndev, "device_of_rank0": dev,
                       "final_rmse_px": round(rmse, 6), "initial_rmse_px": round(float(np.sqrt(out['initial_sse'] / n_obs_total)), 6),
                       "pcg_iterations_per_lm": round(out["pcg_iterations"] / max(steps_done, 1), 2),
                       "accepted_steps": out["accepted"],
                       "seconds": {k: round(out[k], 6) for k in ("seconds_total", "seconds_linearize", "seconds_pcg", "seconds_update")}},
            "roofline": roofline,
            **({"weak_scaling": weak} if weak is not None else {}),
            "kernel_profile_us": {k: round(v["working_mean_us"], 3) for k, v in prof.items()},
        }
        if cpu is not None:
            ng = cpu.pop("n_groups")
            line["cpu_baseline"] = cpu
            if intr0 is None:
                line["cpu_baseline_vectorised"] = cpu_baseline_vectorised(prob, ng)
        print(json.dumps(line), flush=True)
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
