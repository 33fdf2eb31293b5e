"""Two-level preconditioner for band-structured problems (BASELINE config 5's topology; csrc/ba_coarse.hpp).
CPU: the oracle's block-assembled coarse matrix equals P^T S P obtained by applying the matrix-free operator to the
coarse basis, and the additive coarse correction cuts the PCG iteration count of a chain problem.  GPU: the device
solve with the coarse level against the oracle's mirror (same preconditioner): PCG iteration counts per LM iteration
and the iterates; against the single-level device solve: same minimiser, far fewer PCG iterations."""
import numpy as np
import pytest

from bundle_adjustment_amd.synthetic import make_bal_like, make_problem
from oracle import ba_oracle as o


def _chain(n_cams=120, seed=3):
    scale = n_cams / 1723.0
    return make_bal_like(n_cams=n_cams, n_pts=int(156502 * scale), n_obs_target=int(678718 * scale), seed=seed)


def test_oracle_coarse_matrix_is_the_galerkin_product():
    p = _chain(70)
    ne = o.normal_equations(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber")
    op = o.SchurOperator(ne, p.cam_idx, p.pt_idx, 1e-3, 0)
    E = o.coarse_matrix(op, 0)
    nc, na = p.n_cams, (p.n_cams + o.AGG - 1) // o.AGG
    P = np.zeros((nc, 6, na, 6))
    for c in range(1, nc):
        for d in range(6):
            P[c, d, c // o.AGG, d] = 1.0
    P = P.reshape(6 * nc, 6 * na)
    SP = np.stack([op.apply(P[:, j].reshape(nc, 6)).ravel() for j in range(6 * na)], axis=1)
    G = P.T @ SP
    assert np.abs(E - G).max() <= 1e-9 * np.abs(G).max()
    assert np.all(np.linalg.eigvalsh(0.5 * (E + E.T)) > 0)


def test_coarse_correction_cuts_pcg_iterations_on_a_chain():
    p = _chain(160)
    out = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber", max_iters=4, ftol=0, xtol=0, gtol=0, pcg_tol=0.1,
                     pcg_max_iters=500)
    ne = o.normal_equations(out["cams"], out["pts"], p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber")
    op = o.SchurOperator(ne, p.cam_idx, p.pt_idx, 1e-4, 0)
    Minv = np.linalg.inv(op.schur_diag_blocks())
    rhs = op.rhs()
    _, it1, _ = o.pcg(op, rhs, Minv, 0.05, 2000)
    M2 = o.two_level_apply(Minv, np.linalg.inv(o.coarse_matrix(op, 0)), 0)
    x2, it2, _ = o.pcg(op, rhs, M2, 0.05, 2000)
    assert it2 < 0.6 * it1, (it1, it2)
    x1, _, _ = o.pcg(op, rhs, Minv, 1e-10, 5000)
    x2, _, _ = o.pcg(op, rhs, M2, 1e-10, 5000)
    assert np.abs(x1 - x2).max() <= 1e-6 * np.abs(x1).max()                 # the same solution of S x = g


@pytest.mark.gpu
def test_device_two_level_matches_the_oracle_mirror_and_beats_single_level():
    from bundle_adjustment_amd import hip_backend
    p = _chain(200, seed=5)
    kw = dict(loss="huber", max_iters=6, ftol=0.0, xtol=0.0, gtol=0.0, pcg_tol=0.1, pcg_max_iters=600)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        two = s.solve(preconditioner="two_level", **kw)
        tr2 = s.trace()
        cams2, pts2 = s.get_params()
        s.set_problem(p)
        one = s.solve(preconditioner="schur_jacobi", **kw)
        tr1 = s.trace()
    ref = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber", max_iters=6, ftol=0, xtol=0, gtol=0, pcg_tol=0.1,
                     pcg_max_iters=600, precond="two_level")
    # same preconditioner, same recurrences: PCG iteration counts agree with the CPU mirror (a count may differ by one
    # where the stopping ratio sits on the threshold), and so do the iterates
    its_dev = [t["pcg_iterations"] for t in tr2]
    its_ref = [hrec["pcg"] for hrec in ref["history"]]
    assert len(its_dev) == len(its_ref) and max(abs(a - b) for a, b in zip(its_dev, its_ref)) <= 2, (its_dev, its_ref)
    assert abs(two["final_cost"] - ref["cost"]) <= 1e-6 * ref["cost"]
    # against the single level: fewer PCG iterations for the same descent (13 aggregates only here; ~3x on C5)
    assert two["pcg_iterations"] < 0.8 * one["pcg_iterations"], (two["pcg_iterations"], one["pcg_iterations"])
    assert two["final_cost"] <= one["final_cost"] * 1.02
    # bit-reproducible (fixed-point accumulation of the coarse matrix)
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        again = s.solve(preconditioner="two_level", **kw)
        cams3, pts3 = s.get_params()
    assert again["final_cost"] == two["final_cost"] and np.array_equal(cams2, cams3) and np.array_equal(pts2, pts3)


@pytest.mark.gpu
def test_two_level_is_refused_where_the_problem_has_no_band_structure():
    from bundle_adjustment_amd import hip_backend
    p = make_problem(40, 3000, 6, seed=2)                                   # every point seen from random cameras
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        with pytest.raises(hip_backend.BAHipError, match="band-structured"):
            s.solve(preconditioner="two_level", max_iters=3)
        out = s.solve(max_iters=5)                                          # default: Schur-Jacobi
        assert out["final_cost"] < out["initial_cost"]
