"""GPU property tests (hypothesis): random small problems through the C ABI against the oracle
and against size-independent invariants -- observation permutation, duplicated observations,
points seen by a single camera, rigid change of the world frame."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.problem import BAProblem
from bundle_adjustment_amd.synthetic import make_problem
from oracle import ba_oracle as o

pytestmark = pytest.mark.gpu
# BA_HYPOTHESIS_EXAMPLES raises the example count for a stress run (default: a quick 12 per property)
COMMON = dict(deadline=None, max_examples=int(os.environ.get("BA_HYPOTHESIS_EXAMPLES", "12")), suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def solver():
    s = hip_backend.Solver(0)
    yield s
    s.close()


def _random_problem(seed, n_cams, n_pts, k, outliers):
    return make_problem(n_cams, n_pts, min(k, n_cams), seed=seed, outlier_frac=outliers)


@settings(**COMMON)
@given(seed=st.integers(0, 10_000), n_cams=st.integers(2, 9), n_pts=st.integers(5, 120), k=st.integers(1, 5),
       loss=st.sampled_from(["linear", "huber"]))
def test_kernels_match_oracle_on_random_problems(solver, seed, n_cams, n_pts, k, loss):
    p = _random_problem(seed, n_cams, n_pts, k, 0.05)
    solver.set_problem(p)
    r, sse, cost = solver.residuals(loss)
    ref = o.residuals(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4)
    assert np.abs(r - ref).max() <= 1e-9
    assert abs(cost - o.robust_cost(ref, loss)) <= 1e-10 * max(1.0, cost)
    Hcc, bc, Hpp, bp = solver.linearize(loss)
    ne = o.normal_equations(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, loss)
    for a, b in ((Hcc, o.sym6_pack(ne["Hcc"])), (bc, ne["bc"]), (Hpp, o.sym3_pack(ne["Hpp"])), (bp, ne["bp"])):
        assert np.abs(a - b).max() <= 1e-9 * max(1e-300, np.abs(b).max())
    lam = 1e-2
    op = o.SchurOperator(ne, p.cam_idx, p.pt_idx, lam, 0)
    v = np.random.default_rng(seed).normal(size=(p.n_cams, 6))
    v[0] = 0
    sv = solver.schur_apply(lam, v)
    assert np.abs(sv - op.apply(v)).max() <= 1e-9 * np.abs(op.apply(v)).max()
    g = solver.schur_rhs(lam)
    assert np.abs(g - op.rhs()).max() <= 1e-9 * max(1e-300, np.abs(op.rhs()).max())


@settings(**COMMON)
@given(seed=st.integers(0, 10_000), n_cams=st.integers(3, 8), n_pts=st.integers(20, 150))
def test_permutation_and_duplicates(solver, seed, n_cams, n_pts):
    p = _random_problem(seed, n_cams, n_pts, 3, 0.0)
    rng = np.random.default_rng(seed)
    solver.set_problem(p)
    r0, sse0, _ = solver.residuals()
    H0 = solver.linearize("huber")
    perm = rng.permutation(p.n_obs)
    q = BAProblem(p.cams, p.pts, p.cam_idx[perm], p.pt_idx[perm], p.uv[perm], p.K4, 0)
    solver.set_problem(q)
    r1, sse1, _ = solver.residuals()
    assert np.array_equal(r1, r0[perm])                       # rows follow the caller's order
    H1 = solver.linearize("huber")
    for a, b in zip(H0, H1):                                   # sums differ only by order of addition
        assert np.abs(a - b).max() <= 1e-11 * max(1e-300, np.abs(a).max())
    # a duplicated observation counts twice (the reference keeps both rows, src/bundle_adjuster.py:213-216)
    d = BAProblem(p.cams, p.pts, np.r_[p.cam_idx, p.cam_idx[:1]], np.r_[p.pt_idx, p.pt_idx[:1]],
                  np.r_[p.uv, p.uv[:1]], p.K4, 0)
    solver.set_problem(d)
    r2, sse2, _ = solver.residuals()
    assert np.array_equal(r2[:-1], r0) and np.array_equal(r2[-1], r0[0])
    assert abs(sse2 - (sse0 + float((r0[0] ** 2).sum()))) <= 1e-10 * sse2


@settings(**COMMON)
@given(seed=st.integers(0, 10_000))
def test_solve_handles_single_view_points_and_is_gauge_consistent(solver, seed):
    p = _random_problem(seed, 6, 80, 3, 0.0)
    keep = np.ones(p.n_obs, dtype=bool)
    for j in (3, 11, 17):                                      # these points keep one observation only
        idx = np.nonzero(p.pt_idx == j)[0]
        keep[idx[1:]] = False
    q = BAProblem(p.cams, p.pts, p.cam_idx[keep], p.pt_idx[keep], p.uv[keep], p.K4, 0)
    ref = o.lm_solve(q.cams, q.pts, q.cam_idx, q.pt_idx, q.uv, q.K4, 0, "huber", max_iters=30, ftol=1e-12,
                     xtol=1e-12, gtol=0.0, pcg_tol=1e-3)
    for small_solver in (1, 0):                                # the multi-kernel loop, then the single-launch window solver
        solver.set_problem(q)
        out = solver.solve(loss="huber", max_iters=30, ftol=1e-12, xtol=1e-12, gtol=0.0, pcg_tol=1e-3, small_solver=small_solver)
        assert (out["pcg_iterations"] > 0) == bool(small_solver)
        assert np.isfinite(out["final_cost"]) and out["final_cost"] <= out["initial_cost"]
        cams, pts = solver.get_params()
        assert np.all(np.isfinite(cams)) and np.all(np.isfinite(pts))
        np.testing.assert_array_equal(cams[0], q.cams[0])          # the fixed camera never moves
        assert abs(out["final_cost"] - ref["cost"]) <= 1e-7 * max(1.0, ref["cost"])
