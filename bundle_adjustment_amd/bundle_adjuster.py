"""Drop-in for the reference's ``BundleAdjuster`` (``src/bundle_adjuster.py:16-240``).

Same constructor, public attributes (``camera_matrix``, ``window_size`` -- callers mutate
``window_size`` for the final global BA, ``src/main.py:83-86``), same ``run(gmap)`` control
flow, skip / divergence behaviour, map write-back shapes and log lines (parsed by
``src/analyze_log.py:42-45``), same private helper names and signatures.  The solve itself
(``scipy.optimize.least_squares`` at ``:170-174`` and everything it calls back) is replaced
by the HIP Levenberg-Marquardt / Schur / PCG solver behind ``libba_hip.so``; there is no CPU
path in this class.

Differences, by design:
* the Jacobian is analytic on the device, so ``_prepare_sparsity_matrix`` is kept only for
  API compatibility (it is not needed by ``run``) and the spy-plot step (``:168``) is an
  optional hook (``sparsity_plot_hook``), off by default;
* the intermediate ``.pcd`` snapshot (``:187-193``) is written as a plain ASCII PCD, only
  when the ``DEBUG_DIRS['lba_steps']`` directory exists (no open3d dependency);
* extra keyword arguments select solver options; their defaults are the reference's
  literals (``loss='huber'``, ``xtol = ftol = 1e-5``, at most 50 evaluations);
* ``inplace_writeback=True`` (opt-in) stores optimised landmark positions into their existing arrays instead of
  rebinding ``MapPoint.position`` as the reference does (``:239-240``): faster on 100 k landmarks, but aliases of
  the old array then see the new values;
* ``comm=(rank, world, unique_id)`` makes ``run`` SPMD over ``world`` processes, one GPU each
  (SURVEY.md section 8e): every rank calls ``run`` with an identical map, solves its landmark shard
  (the library all-reduces the reduced camera system), gathers all points and writes the whole map
  back, so the ranks' maps stay identical.  ``unique_id`` = ``hip_backend.comm_unique_id()`` of
  rank 0, shipped by the caller (any transport).
"""
from __future__ import annotations

import os

import numpy as np

from . import hip_backend
from .map_structures import Map
from .parameters import DEBUG_DIRS
from .problem import BAProblem, WindowCache, extract_shard, flatten_map_window_ids, gather_window, shard_by_landmark
from .rotations import matrices_to_rvecs


class BundleAdjuster:
    def __init__(self, camera_matrix, window_size=5, *, device_id=0, loss='huber', f_scale=1.0, ftol=1e-5,
                 xtol=1e-5, gtol=1e-8, max_iters=50, pcg_tol=0.1, pcg_max_iters=200, pcg_model_tol=0.0, preconditioner='schur_jacobi',
                 jacobian='f64', comm=None, sparsity_plot_hook=None, verbose=0, reuse_window=True, metrics_path=None,
                 inplace_writeback=False, reuse_min_obs=20000):
        self.camera_matrix = camera_matrix
        self.window_size = window_size
        self.device_id = device_id
        self.solver_options = dict(loss=loss, f_scale=f_scale, ftol=ftol, xtol=xtol, gtol=gtol, max_iters=max_iters,
                                   pcg_tol=pcg_tol, pcg_max_iters=pcg_max_iters, pcg_model_tol=pcg_model_tol, preconditioner=preconditioner,
                                   jacobian_precision={'f64': 0, 'f32': 1}[jacobian], verbose=verbose)
        self.comm = comm                       # None, or (rank, world, unique_id bytes)
        self.sparsity_plot_hook = sparsity_plot_hook
        self.last_summary = None
        self.metrics_path = metrics_path       # JSON lines, one per run(): sizes, summary, per-iteration trace
        # False (default): landmark positions are REBOUND to fresh (3,1) arrays, as the reference does
        # (src/bundle_adjuster.py:239-240) -- whoever holds the previous array keeps the previous values.
        # True: the numbers are stored into the existing arrays (no object per landmark; aliases see the update).
        self.inplace_writeback = inplace_writeback
        self._solver = None
        # what survives between consecutive run() calls (src/pipeline.py:99 calls run after every keyframe): the
        # flattened window (problem.WindowCache) and, while its observation structure is unchanged, the problem the
        # solver already holds on the device
        # (windows with fewer observation-list entries than reuse_min_obs are walked afresh on every call: problem.WindowCache)
        self._window = WindowCache(min_obs=reuse_min_obs) if reuse_window else None
        self._uploaded_token = None

    # -- device -------------------------------------------------------------------------
    def _get_solver(self):
        if self._solver is None:
            self._solver = hip_backend.Solver(self.device_id)
            if self.comm is not None and self.comm[1] > 1:
                rank, world, unique_id = self.comm
                self._solver.comm_init(rank, world, unique_id)
        return self._solver

    def _write_metrics(self, prob, local_kf_ids, summary, solver):
        trace = solver.trace() if hasattr(solver, "trace") else []
        with open(self.metrics_path, "a") as f:
            f.write(_metrics_line(prob, local_kf_ids, summary, trace) + "\n")

    def close(self):
        if self._solver is not None:
            self._solver.close()
            self._solver = None
        self._uploaded_token = None

    # -- reference-compatible helpers -----------------------------------------------------
    def _cost_function(self, params, fixed_kf_pose, fixed_kf_id, adjustable_kf_ids, map_point_ids, observations,
                       keypoints_2d):
        """Reprojection residuals (observed - projected, x then y per observation, rows in
        ``observations`` order) -- ``src/bundle_adjuster.py:24-72`` evaluated on the GPU."""
        na, npnt = len(adjustable_kf_ids), len(map_point_ids)
        params = np.asarray(params, dtype=np.float64)
        kf_index = {fixed_kf_id: 0}
        kf_index.update({k: i + 1 for i, k in enumerate(adjustable_kf_ids)})
        mp_index = {m: i for i, m in enumerate(map_point_ids)}
        # rows the reference skips (unknown map point / keyframe, :54,:63) are dropped here too
        kept = [(kf_index[k], mp_index[m], keypoints_2d[(k, m)]) for k, m in observations
                if m in mp_index and k in kf_index]
        fixed_R, fixed_t = fixed_kf_pose
        cams = np.empty((na + 1, 6))
        cams[0, :3] = matrices_to_rvecs(np.asarray(fixed_R, dtype=np.float64)[None])[0]
        cams[0, 3:] = np.asarray(fixed_t, dtype=np.float64).ravel()
        cams[1:, :3] = params[:3 * na].reshape(na, 3)
        cams[1:, 3:] = params[3 * na:6 * na].reshape(na, 3)
        pts = params[6 * na:].reshape(npnt, 3)
        K = np.asarray(self.camera_matrix, dtype=np.float64)
        prob = BAProblem(cams, pts, np.array([a for a, _, _ in kept], dtype=np.int32),
                         np.array([b for _, b, _ in kept], dtype=np.int32),
                         np.array([c for _, _, c in kept], dtype=np.float64).reshape(len(kept), 2),
                         np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), 0)
        s = self._get_solver()
        s.set_problem(prob)
        self._uploaded_token = None               # the solver no longer holds run()'s window
        r, _, _ = s.residuals('linear')
        return r.ravel()

    def _prepare_sparsity_matrix(self, num_adj_kfs, num_mps, adj_kf_ids, mp_ids, observations):
        """0/1 structure of the Jacobian the reference hands to scipy
        (``src/bundle_adjuster.py:74-120``): rows 2i, 2i+1 <-> observation i; columns
        [rvec(3 Na) | tvec(3 Na) | points(3 Np)].  Not used by ``run`` (analytic Jacobian)."""
        from scipy.sparse import lil_matrix
        A = lil_matrix((len(observations) * 2, num_adj_kfs * 6 + num_mps * 3), dtype=int)
        adj = {kf: i for i, kf in enumerate(adj_kf_ids)}
        mpi = {mp: i for i, mp in enumerate(mp_ids)}
        for i, (kf, mp) in enumerate(observations):
            m = mpi.get(mp)
            if m is None:
                continue
            rows = slice(2 * i, 2 * i + 2)
            A[rows, num_adj_kfs * 6 + 3 * m:num_adj_kfs * 6 + 3 * m + 3] = 1
            k = adj.get(kf)
            if k is not None:
                A[rows, 3 * k:3 * k + 3] = 1
                A[rows, 3 * num_adj_kfs + 3 * k:3 * num_adj_kfs + 3 * k + 3] = 1
        return A

    def _gather_local_data(self, gmap: Map, local_kf_ids: list):
        """``src/bundle_adjuster.py:195-218``."""
        return gather_window(gmap, local_kf_ids)

    def _update_map(self, gmap: Map, optimized_params: np.ndarray, adjustable_kf_ids: list,
                    local_map_point_ids: list, rotations=None):
        """Write optimised poses / points back in place (``src/bundle_adjuster.py:220-240``):
        ``R`` (3,3), ``t`` (3,1), ``position`` (3,1).  ``rotations`` (Na,3,3), when given,
        are the device's Rodrigues matrices of the optimised rotation vectors."""
        na = len(adjustable_kf_ids)
        x = np.asarray(optimized_params, dtype=np.float64)
        rvecs = x[:3 * na].reshape(na, 3)
        tvecs = x[3 * na:6 * na].reshape(na, 3)
        pts = x[6 * na:].reshape(len(local_map_point_ids), 3)
        if rotations is None:
            from .rotations import rvecs_to_matrices
            rotations = rvecs_to_matrices(rvecs)
        keyframes, map_points = gmap.keyframes, gmap.map_points
        for i, kf_id in enumerate(adjustable_kf_ids):
            keyframes[kf_id].R = np.array(rotations[i], dtype=np.float64).reshape(3, 3)
            keyframes[kf_id].t = tvecs[i].reshape(3, 1).copy()
        # Points.  Default: rebind every landmark's position to a fresh (3,1) view into the result array, exactly the
        # reference's `position = p.reshape(3, 1)` (:239-240), in one native loop (csrc/mapwalk.c rebind_positions).
        # inplace_writeback=True: store the three numbers into the landmark's EXISTING (3,1) float64 array instead
        # (scatter_positions; no Python object per landmark) -- anything that aliases that array then sees the update,
        # which the reference's rebinding does not do; landmarks whose position is anything else are rebound.
        from . import _mapwalk
        ids = np.ascontiguousarray(local_map_point_ids, dtype=np.int64)
        pts = np.ascontiguousarray(pts)
        views = pts.reshape(-1, 3, 1)
        if not self.inplace_writeback:
            _mapwalk.rebind_positions(map_points, ids, views)
            return
        todo = _mapwalk.scatter_positions(map_points, ids, pts)
        for i in todo:
            map_points[int(ids[i])].position = views[i]

    # -- the solve step -------------------------------------------------------------------
    def run(self, gmap: Map):
        """Sliding-window / global bundle adjustment, ``src/bundle_adjuster.py:122-193``."""
        print("    --- Running Local Bundle Adjustment ---")
        all_kf_ids = sorted(gmap.keyframes.keys())
        if len(all_kf_ids) < self.window_size:
            print("    -> LBA Skipped: Not enough keyframes.")
            return
        local_kf_ids = all_kf_ids[-(self.window_size + 1):-1]      # newest keyframe excluded (:139)
        fixed_kf_id = local_kf_ids[0]
        adjustable_kf_ids = local_kf_ids[1:]
        if not adjustable_kf_ids:
            print("    -> LBA Skipped: No adjustable keyframes.")
            return
        # array-level form of _gather_local_data + the parameter packing of :157-162
        if self._window is not None:
            prob, local_map_point_ids, token = self._window.flatten(gmap, local_kf_ids, self.camera_matrix)
        else:
            prob, local_map_point_ids = flatten_map_window_ids(gmap, local_kf_ids, self.camera_matrix)
            token = None
        if len(local_map_point_ids) == 0:
            print("    -> LBA Skipped: No points in the local window.")
            return

        if self.sparsity_plot_hook is not None:
            _, observations, _ = self._gather_local_data(gmap, local_kf_ids)
            self.sparsity_plot_hook(self._prepare_sparsity_matrix(len(adjustable_kf_ids), len(local_map_point_ids),
                                                                  adjustable_kf_ids, local_map_point_ids.tolist(), observations),
                                    fixed_kf_id, local_kf_ids[-1])
        solver = self._get_solver()
        world = self.comm[1] if self.comm is not None else 1
        # same observation structure as the problem the solver already holds (a repeated run on an unchanged window,
        # e.g. the final global BA after the last sliding-window one): parameters only, no re-sort, no re-upload
        same_structure = token is not None and token == self._uploaded_token
        if world > 1:                          # this rank's landmark block; all cameras
            if prob.n_pts < world:             # (the same on every rank: nobody enters a collective)
                raise ValueError(f"{prob.n_pts} landmarks in the window cannot be sharded over {world} ranks; "
                                 f"run this window on a single rank")
            p_begin, p_end = shard_by_landmark(prob, world)[self.comm[0]]
            if same_structure:
                solver.set_params(prob.cams, prob.pts[p_begin:p_end])
            else:
                shard, _ = extract_shard(prob, p_begin, p_end)
                solver.set_problem(shard)
        elif same_structure:
            solver.set_params(prob.cams, prob.pts)
        else:
            solver.set_problem(prob)
        self._uploaded_token = token
        summary = solver.solve(**self.solver_options)          # costs / verdicts are global on every rank
        self.last_summary = summary
        if self.metrics_path and (self.comm is None or self.comm[0] == 0):
            self._write_metrics(prob, local_kf_ids, summary, solver)
        initial_cost, final_cost = summary["initial_sse"], summary["final_sse"]     # plain SSE (:165, :176)
        if final_cost >= initial_cost:
            print(f"    -> LBA Diverged! Cost increased from {initial_cost:.2f} to {final_cost:.2f}. Discarding results.")
            return

        cams, pts = solver.get_params()
        if world > 1:
            pts = solver.allgather_points(p_begin, prob.n_pts)
        R = solver.get_rotations()
        x = np.concatenate([cams[1:, :3].ravel(), cams[1:, 3:].ravel(), pts.ravel()])
        self._update_map(gmap, x, adjustable_kf_ids, local_map_point_ids, rotations=R[1:])

        improvement = 100.0 * (initial_cost - final_cost) / (initial_cost + 1e-8)
        print(f"    -> LBA Complete. Initial Cost: {initial_cost:.2f}, Final Cost: {final_cost:.2f}, "
              f"Improvement: {improvement:.2f}%")

        lba_steps_dir = DEBUG_DIRS['lba_steps']
        if os.path.isdir(lba_steps_dir):
            pcd = gmap.get_pcd()
            if pcd.has_points():
                pcd_filename = os.path.join(lba_steps_dir, f"map_after_lba_kf_{fixed_kf_id}.pcd")
                _write_ascii_pcd(pcd_filename, pcd.points, pcd.colors)
                print(f"    -> Saved intermediate map to {pcd_filename}")


def _metrics_line(prob, local_kf_ids, summary, trace):
    """One JSON object per run(): what the reference only prints as a log line (src/bundle_adjuster.py:183-184) plus
    the trajectory -- sizes, costs, iterations, seconds, and one record per LM iteration (ba_get_trace)."""
    import json
    rec = dict(event="lba", fixed_keyframe=int(local_kf_ids[0]), last_keyframe=int(local_kf_ids[-1]),
               n_keyframes=int(prob.n_cams), n_landmarks=int(prob.n_pts), n_observations=int(prob.n_obs))
    rec.update({k: (float(v) if isinstance(v, float) else v) for k, v in summary.items()})
    n_obs = max(int(prob.n_obs), 1)
    rec["initial_rmse_px"] = float(np.sqrt(summary["initial_sse"] / n_obs))
    rec["final_rmse_px"] = float(np.sqrt(summary["final_sse"] / n_obs))
    rec["lm_iterations_per_s"] = float(summary["iterations"] / summary["seconds_total"]) if summary.get("seconds_total") else None
    rec["trace"] = trace
    return json.dumps(rec)


def _write_ascii_pcd(path, points, colors):
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    cols = np.clip(np.asarray(colors, dtype=np.float64).reshape(-1, 3), 0.0, 1.0)
    rgb = (np.round(cols * 255).astype(np.uint32) @ np.array([65536, 256, 1], dtype=np.uint32)).astype(np.uint32)
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\n"
                "TYPE F F F U\nCOUNT 1 1 1 1\n")
        f.write(f"WIDTH {pts.shape[0]}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {pts.shape[0]}\nDATA ascii\n")
        for p, c in zip(pts, rgb):
            f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g} {int(c)}\n")
