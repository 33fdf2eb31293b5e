"""CPU: the oracle's analytic Jacobian blocks (what the device kernels are checked against)
versus finite differences of the oracle residual, which is itself pinned to the reference's
_cost_function vectors -- including theta = 0, theta ~ pi, a tiny theta, a non-orthogonal R
and a point behind the cameras (golden cost_edge) -- and versus scipy's own
approx_derivative (scipy/optimize/_numdiff.py:277), the routine the reference relies on."""
import numpy as np
import pytest
from scipy.optimize._numdiff import approx_derivative

from oracle import ba_oracle as o
from tests.helpers import golden_flat_problem, load_golden


def _fd_blocks(p, i, h=1e-6):
    c, q = int(p.cam_idx[i]), int(p.pt_idx[i])
    sl = slice(i, i + 1)

    def res(cams, pts):
        return o.residuals(cams, pts, p.cam_idx[sl], p.pt_idx[sl], p.uv[sl], p.K4)[0]
    Jc, Jp = np.zeros((2, 6)), np.zeros((2, 3))
    for j in range(6):
        a, b = p.cams.copy(), p.cams.copy()
        a[c, j] += h
        b[c, j] -= h
        Jc[:, j] = (res(a, p.pts) - res(b, p.pts)) / (2 * h)
    for j in range(3):
        a, b = p.pts.copy(), p.pts.copy()
        a[q, j] += h
        b[q, j] -= h
        Jp[:, j] = (res(p.cams, a) - res(p.cams, b)) / (2 * h)
    return Jc, Jp


@pytest.mark.parametrize("name", ["cost_edge", "cost_seed0"])
def test_analytic_blocks_match_central_differences(name):
    p = golden_flat_problem(load_golden(name))
    Jc, Jp = o.jacobian_blocks(p.cams, p.pts, p.cam_idx, p.pt_idx, p.K4)
    idx = range(p.n_obs) if p.n_obs < 400 else range(0, p.n_obs, 7)
    worst = 0.0
    for i in idx:
        a, b = _fd_blocks(p, i)
        worst = max(worst, np.abs(a - Jc[i]).max() / max(1.0, np.abs(a).max()),
                    np.abs(b - Jp[i]).max() / max(1.0, np.abs(b).max()))
    assert worst <= 5e-8, worst
    thetas = np.linalg.norm(p.cams[:, :3], axis=1)
    if name == "cost_edge":
        assert thetas.min() == 0.0 and thetas.max() > 3.1       # the special branches are really in there


def test_sparse_jacobian_matches_scipy_approx_derivative():
    p = golden_flat_problem(load_golden("cost_seed1"))
    x0, _ = o.pack_reference_params(p.cams, p.pts, 0)
    fun = o.flat_residual_fun(p.cams, p.n_pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0)
    J = o.flat_jacobian_fun(p.cams, p.n_pts, p.cam_idx, p.pt_idx, p.K4, 0)(x0)
    sparsity = o.flat_sparsity(p.n_cams, p.n_pts, p.cam_idx, p.pt_idx, 0)
    Jfd = approx_derivative(fun, x0, method="3-point", sparsity=sparsity)
    assert abs(J - Jfd).max() <= 1e-5 * abs(J).max()
    assert (J != 0).sum() <= sparsity.nnz                       # analytic blocks live inside the reference's pattern


def test_right_jacobian_series_is_continuous():
    axis = np.array([0.3, -0.5, 0.81])
    axis /= np.linalg.norm(axis)
    for t in (0.0499, 0.0501, 1e-9, 1e-4):
        M = o.so3_right_jacobian((axis * t)[None])[0]
        b, d = (1 - np.cos(t)) / t**2 if t > 1e-6 else 0.5, (t - np.sin(t)) / t**3 if t > 1e-3 else 1 / 6
        rx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]]) * t
        np.testing.assert_allclose(M, np.eye(3) - b * rx + d * rx @ rx, atol=1e-9)


def test_dense_and_matrix_free_reduced_systems_give_the_same_lm_trajectory():
    """oracle lm_solve(linear_solver='dense') -- the checker of the single-launch window solver (csrc/ba_small.hpp) --
    against the same loop with the matrix-free operator and PCG driven to round-off: the explicit Schur complement and
    its right-hand side are the operator's, so every trial cost agrees."""
    from bundle_adjustment_amd.synthetic import make_problem
    p = make_problem(4, 60, 3, seed=1, outlier_frac=0.05)
    kw = dict(fixed_cam=0, loss="huber", max_iters=5, ftol=0.0, xtol=0.0, gtol=0.0)
    a = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, linear_solver="dense", **kw)
    b = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, pcg_tol=1e-13, pcg_max_iters=2000, pcg_model_tol=0.0, **kw)
    assert a["pcg_iters"] == 0 and b["pcg_iters"] > 0
    for ha, hb in zip(a["history"], b["history"]):
        assert abs(ha["cost_new"] - hb["cost_new"]) <= 1e-10 * hb["cost_new"]
        assert abs(ha["step"] - hb["step"]) <= 1e-8 * hb["step"]
