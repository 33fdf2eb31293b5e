#!/usr/bin/env python3
"""Converged-solution goldens at BASELINE sizes: scipy's own ``least_squares`` (TRF) driven to
convergence on the REFERENCE'S residual function, so that the north star's "final reprojection
RMSE within 1e-6 of reference" is pinned at C2 (50 cams / 5k pts / 30k obs) and C3 (1000 / 100k /
1M), not only on the 4-keyframe scenes of ``make_golden.py`` section D.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_converged.py --config C2 --loss linear
    python tests/golden/make_golden_converged.py --config C2 --loss huber --warm-start --max-nfev 3000
    python tests/golden/make_golden_converged.py --config C3 --loss linear
    python tests/golden/make_golden_converged.py --config C3 --loss huber --warm-start --skip-fd-check
    python tests/golden/make_golden_converged.py --config C2 --loss huber --certify gpurun_out/xstar_c2_huber.npy
    python tests/golden/make_golden_converged.py --config C3 --loss huber --certify gpurun_out/xstar_c3_huber.npy --fun oracle

What runs (match: ``/root/reference/src/bundle_adjuster.py:170-176``):

* ``fun`` = the imported reference's ``BundleAdjuster._cost_function`` (``--fun reference``:
  its per-observation loop, unmodified, loaded exactly as ``make_golden.py`` loads it), or the
  oracle's vectorised restatement of it (``--fun oracle``; C3 only, where one sweep of the
  reference's loop takes ~35 s).  In the second case the reference's own function is still
  evaluated ONCE at the start point and ONCE at the solution scipy returns, and the stored SSE /
  RMSE / cost are the ones computed from the reference's residual vector.
* ``jac`` = the oracle's analytic CSR Jacobian (``oracle.ba_oracle.flat_jacobian_fun``), cross-checked
  at x0 against scipy's own 2-point finite differences of ``fun`` over the reference's sparsity
  colour groups (the path ``least_squares`` takes for the reference: ``jac_sparsity=A``).
* ``scipy.optimize.least_squares(fun, x0, jac, method='trf', tr_solver='lsmr', x_scale='jac',
  loss=<linear|huber>, f_scale=1)`` with tolerances far below the reference's 1e-5, so that it stops
  at the minimum instead of after 4-5 iterations.

``--certify <x*.npy>`` (the Huber pins as CERTIFICATES instead of stalled scipy crawls): x* is the minimiser the device
solver exported (``tools/export_converged.py``, run on the GPU box) in the reference's parameter packing.  In the build
container the imported reference's ``_cost_function`` is evaluated AT x*, the Huber gradient ``J^T (rho' f)`` is formed with
the oracle's analytic CSR Jacobian and its max-norm recorded, and scipy's ``least_squares`` is restarted FROM x* with the
reference's kind of tolerances: it must stop on gtol / ftol / xtol within a few evaluations without lowering the cost by more
than 1e-9 relative.  Written to ``cert_<config>_huber.npz`` (C2: with x* itself; C3: its SHA-256 and the scalars).

The synthetic problem is regenerated from ``(config, seed)`` by ``bundle_adjustment_amd.synthetic``;
the fixture stores only scalars plus a checksum of the inputs, so the tests notice if the generator
ever changes.  Same cv2 caveat as every golden here: parity unpinned at the cv2 boundary.
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def problem_checksum(p):
    h = hashlib.sha256()
    for a in (p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=["C2", "C3"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--loss", default="linear", choices=["linear", "huber"])
    ap.add_argument("--fun", default="reference", choices=["reference", "oracle"])
    ap.add_argument("--max-nfev", type=int, default=400)
    ap.add_argument("--lsmr-tol", type=float, default=1e-10)
    ap.add_argument("--skip-fd-check", action="store_true")
    ap.add_argument("--chunk", type=int, default=200, help="evaluations per scipy call; the fixture is rewritten after each")
    ap.add_argument("--stall", type=float, default=2e-8, help="stop when the RMSE moved less than this over each of the last two chunks")
    ap.add_argument("--certify", default=None, metavar="XSTAR.npy",
                    help="certify a minimiser exported by tools/export_converged.py instead of running scipy from the start")
    ap.add_argument("--warm-start", action="store_true",
                    help="first run the same scipy call with loss='linear' to convergence and start the requested loss there "
                         "(scipy's Huber TRF crawls for hundreds of iterations from a 9 px start; the minimum pinned is the same)")
    args = ap.parse_args()

    import make_golden as mg
    ba_mod, _ms = mg.load_reference()
    from scipy.optimize import least_squares
    from scipy.optimize._numdiff import approx_derivative, group_columns
    from bundle_adjustment_amd.synthetic import make_config
    from oracle import ba_oracle as o

    p = make_config(args.config, seed=args.seed)
    K = np.array([[p.K4[0], 0.0, p.K4[2]], [0.0, p.K4[1], p.K4[3]], [0.0, 0.0, 1.0]])
    ba = ba_mod.BundleAdjuster(K, window_size=p.n_cams)
    x0, adj = o.pack_reference_params(p.cams, p.pts, p.fixed_cam)
    observations = [(int(c), int(q)) for c, q in zip(p.cam_idx, p.pt_idx)]
    kp2d = {ob: (float(u), float(v)) for ob, (u, v) in zip(observations, p.uv)}
    pose = (o.rodrigues_to_mat(p.cams[p.fixed_cam, :3]), p.cams[p.fixed_cam, 3:].reshape(3, 1))
    mp_ids = list(range(p.n_pts))
    ref_args = (pose, p.fixed_cam, adj, mp_ids, observations, kp2d)

    def ref_fun(x):
        return ba._cost_function(x, *ref_args)

    oracle_fun = o.flat_residual_fun(p.cams, p.n_pts, p.cam_idx, p.pt_idx, p.uv, p.K4, p.fixed_cam)
    jac = o.flat_jacobian_fun(p.cams, p.n_pts, p.cam_idx, p.pt_idx, p.K4, p.fixed_cam)

    if args.certify:
        return certify(args, p, x0, ref_fun, oracle_fun, jac, o)

    t = time.time()
    f0_ref = ref_fun(x0)
    t_sweep = time.time() - t
    f0_or = oracle_fun(x0)
    print(f"{args.config} seed {args.seed}: n={x0.size} m={f0_ref.size}; reference sweep {t_sweep:.1f} s; "
          f"oracle vs reference residual at x0: {np.abs(f0_ref - f0_or).max():.2e} px", flush=True)
    fd_err = np.nan
    if not args.skip_fd_check:
        # the reference's own Jacobian path: 2-point finite differences over colour groups of its 0/1 pattern
        A = o.flat_sparsity(p.n_cams, p.n_pts, p.cam_idx, p.pt_idx, p.fixed_cam)
        groups = group_columns(A)
        fd_fun = ref_fun if args.fun == "reference" else oracle_fun
        t = time.time()
        Jfd = approx_derivative(fd_fun, x0, method="2-point", f0=fd_fun(x0), sparsity=(A, groups)).tocsr()
        Ja = jac(x0)
        d = (Jfd - Ja)
        fd_err = float(np.abs(d.data).max()) if d.nnz else 0.0
        scale = float(np.abs(Ja.data).max())
        print(f"analytic vs finite-difference Jacobian ({int(groups.max()) + 1} groups, {time.time() - t:.0f} s): "
              f"max abs diff {fd_err:.3e} (largest entry {scale:.3e})", flush=True)
        assert fd_err <= 2e-4 * scale, "analytic Jacobian disagrees with the finite-difference path"

    fun = ref_fun if args.fun == "reference" else oracle_fun
    nit = [0]
    t_start = time.time()

    def logged(x):
        f = fun(x)
        nit[0] += 1
        if nit[0] % 5 == 0:
            print(f"  nfev {nit[0]:4d}  sse {float(f @ f):.9f}  rmse {np.sqrt(float(f @ f) / p.n_obs):.9f}  "
                  f"{time.time() - t_start:.0f} s", flush=True)
        return f

    x_start = x0
    if args.warm_start and args.loss != "linear":
        pre = least_squares(logged, x0, jac=jac, method="trf", tr_solver="lsmr", x_scale="jac", loss="linear",
                            xtol=1e-15, ftol=1e-14, gtol=1e-11, max_nfev=args.max_nfev,
                            tr_options=dict(atol=args.lsmr_tol, btol=args.lsmr_tol))
        print(f"warm start: linear-loss solution after {pre.nfev} evaluations, sse {2 * pre.cost:.9f}", flush=True)
        x_start = pre.x
    # the requested loss, in chunks of --chunk evaluations: the fixture is (re)written after every chunk, each chunk
    # restarts scipy from the previous chunk's solution (scipy 1.15 has no callback; a restart only resets the trust
    # radius), so a long crawl can be stopped at any time and still leave a usable pin with its history
    name = f"conv_{args.config.lower()}_{args.loss}.npz"
    history = []            # (cumulative nfev, rmse by the reference's arithmetic) after each chunk
    x_cur, total = x_start, 0
    while total < args.max_nfev:
        res = least_squares(logged, x_cur, jac=jac, method="trf", tr_solver="lsmr", x_scale="jac", loss=args.loss, f_scale=1.0,
                            xtol=1e-15, ftol=1e-14, gtol=1e-11, max_nfev=min(args.chunk, args.max_nfev - total),
                            tr_options=dict(atol=args.lsmr_tol, btol=args.lsmr_tol))
        total += int(res.nfev)
        x_cur = res.x
        f_ref = ref_fun(res.x)                       # the reference's own arithmetic at the solution
        sse = float(f_ref @ f_ref)
        rmse = float(np.sqrt(sse / p.n_obs))
        cost = 0.5 * float(o.huber_rho(f_ref ** 2)[0].sum()) if args.loss == "huber" else 0.5 * sse
        history.append((total, rmse))
        print(f"chunk done: status {res.status} nfev {total} scipy cost {res.cost:.12f} "
              f"reference-evaluated cost {cost:.12f} sse {sse:.9f} rmse {rmse:.9f} optimality {res.optimality:.3e} "
              f"({time.time() - t_start:.0f} s)", flush=True)
        tmp = os.path.join(HERE, name + ".part.npz")     # written beside the fixture, then moved over it: readers never see half a file
        np.savez_compressed(tmp, config=np.array(args.config), seed=args.seed, loss=np.array(args.loss),
                            fun=np.array(args.fun), n_cams=p.n_cams, n_pts=p.n_pts, n_obs=p.n_obs,
                            problem_sha256=np.array(problem_checksum(p)), sse0=float(f0_ref @ f0_ref),
                            res_cost=cost, res_cost_scipy=float(res.cost), res_sse=sse, res_rmse=rmse,
                            res_optimality=float(res.optimality), res_nfev=total, res_status=int(res.status),
                            jac_fd_max_abs_diff=fd_err, lsmr_tol=args.lsmr_tol, warm_start=bool(args.warm_start),
                            rmse_history=np.array(history, dtype=np.float64))
        os.replace(tmp, os.path.join(HERE, name if name.endswith(".npz") else name + ".npz"))
        if res.status in (1, 2, 3, 4):               # gtol / ftol / xtol: scipy itself says converged
            break
        if len(history) >= 3 and abs(history[-1][1] - history[-2][1]) < args.stall and abs(history[-2][1] - history[-3][1]) < args.stall:
            print(f"RMSE moved less than {args.stall:g} px over each of the last two chunks: stopping", flush=True)
            break
    print("wrote", name)


def certify(args, p, x0, ref_fun, oracle_fun, jac, o):
    """Stationarity certificate of a device-exported minimiser (see the module docstring)."""
    from scipy.optimize import least_squares
    assert args.loss == "huber", "the linear-loss pins already are converged scipy runs (status 2)"
    xs = np.load(args.certify)
    assert xs.shape == x0.shape, (xs.shape, x0.shape)
    t = time.time()
    f_ref = ref_fun(xs)                                          # the reference's own arithmetic at x*
    t_sweep = time.time() - t
    f_or = oracle_fun(xs)
    sse = float(f_ref @ f_ref)
    rho, drho, _ = o.huber_rho(f_ref ** 2)
    cost = 0.5 * float(rho.sum())
    J = jac(xs)
    g = J.T @ (drho * f_ref)                                     # gradient of 0.5 sum rho(f^2)
    g_inf = float(np.abs(g).max())
    g0 = J.T @ f_ref
    print(f"{args.config} huber certificate: reference sweep at x* {t_sweep:.1f} s, oracle vs reference residual {np.abs(f_ref - f_or).max():.2e} px; "
          f"cost {cost:.12f} sse {sse:.9f} rmse {np.sqrt(sse / p.n_obs):.9f}; |grad|_inf {g_inf:.3e} (|J^T f|_inf {np.abs(g0).max():.3e})", flush=True)
    fun = ref_fun if args.fun == "reference" else oracle_fun
    # restart scipy FROM x*: the reference's solver (TRF, LSMR; x_scale='jac' so that it can move at all) with scipy's default gtol and
    # tolerances a thousand times tighter than the reference's 1e-5 -- it has to declare convergence at once
    res = least_squares(fun, xs, jac=jac, method="trf", tr_solver="lsmr", x_scale="jac", loss="huber", f_scale=1.0,
                        xtol=1e-10, ftol=1e-10, gtol=1e-8, max_nfev=args.max_nfev if args.max_nfev < 50 else 25,
                        tr_options=dict(atol=args.lsmr_tol, btol=args.lsmr_tol))
    f_end = ref_fun(res.x)
    cost_end = 0.5 * float(o.huber_rho(f_end ** 2)[0].sum())
    rel_drop = (cost - cost_end) / cost
    print(f"scipy from x*: status {res.status} after {res.nfev} evaluations, optimality {res.optimality:.3e}, cost {cost_end:.12f} "
          f"(relative decrease {rel_drop:.3e}), moved |dx|_inf {np.abs(res.x - xs).max():.3e}", flush=True)
    assert res.status in (1, 2, 3, 4), "scipy did not declare convergence when started at x*"
    assert rel_drop <= 1e-9, "scipy lowered the cost from x* by more than 1e-9 relative: x* is not the minimiser"
    out = dict(config=np.array(args.config), seed=args.seed, loss=np.array("huber"), fun=np.array(args.fun),
               n_cams=p.n_cams, n_pts=p.n_pts, n_obs=p.n_obs, problem_sha256=np.array(problem_checksum(p)),
               xstar_sha256=np.array(hashlib.sha256(np.ascontiguousarray(xs).tobytes()).hexdigest()),
               res_cost=cost, res_sse=sse, res_rmse=float(np.sqrt(sse / p.n_obs)), grad_inf=g_inf,
               scipy_status=int(res.status), scipy_nfev=int(res.nfev), scipy_optimality=float(res.optimality),
               scipy_cost_end=cost_end, scipy_relative_decrease=rel_drop, scipy_dx_inf=float(np.abs(res.x - xs).max()))
    if xs.size <= 50_000:
        out["xstar"] = xs
    name = f"cert_{args.config.lower()}_huber.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name)


if __name__ == "__main__":
    main()
