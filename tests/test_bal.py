"""Row f2: BAL text reader / writer and the 9-parameter BAL camera.
CPU: exact round trip through the text format, parser errors, the oracle's analytic 2x9 / 2x3 blocks against central
finite differences, the pinhole conversion.  GPU: the BAL residual kernel (ba_residuals_bal) against the oracle,
<= 1e-9 px, and a BAL problem with shared focal length / no distortion solved through the reference's model."""
import os

import numpy as np
import pytest

from bundle_adjustment_amd.bal import BALProblem, from_pinhole, read_bal, to_pinhole, write_bal
from oracle import ba_oracle as o
from tests.helpers import GOLDEN

TINY = os.path.join(GOLDEN, "tiny_bal.txt")


def test_reader_parses_the_committed_fixture_and_round_trips_exactly(tmp_path):
    p = read_bal(TINY)
    assert (p.n_cams, p.n_pts, p.n_obs) == (6, 40, 160)
    assert p.cams.shape == (6, 9) and p.uv.shape == (160, 2) and p.cam_idx.dtype == np.int32
    r = o.bal_residuals(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv)
    assert 0.3 < np.sqrt((r * r).sum() / p.n_obs) < 1.2          # the fixture's pixels carry 0.5 px noise
    out = tmp_path / "again.txt"
    write_bal(out, p)
    q = read_bal(out)
    for a, b in ((p.cams, q.cams), (p.pts, q.pts), (p.cam_idx, q.cam_idx), (p.pt_idx, q.pt_idx), (p.uv, q.uv)):
        np.testing.assert_array_equal(a, b)
    import gzip
    with open(out, "rb") as f, gzip.open(str(out) + ".gz", "wb") as g:
        g.write(f.read())
    np.testing.assert_array_equal(read_bal(str(out) + ".gz").cams, p.cams)


def test_reader_rejects_malformed_files(tmp_path):
    bad = tmp_path / "bad.txt"
    bad.write_text("2 1 1\n0 0 1.0 2.0\n" + "0.0\n" * 10)                  # too few numbers
    with pytest.raises(ValueError, match="expected"):
        read_bal(bad)
    bad.write_text("1 1 1\n0.5 0 1.0 2.0\n" + "0.0\n" * 12)                # fractional index
    with pytest.raises(ValueError, match="non-integer"):
        read_bal(bad)
    bad.write_text("1 1 1\n3 0 1.0 2.0\n" + "0.0\n" * 12)                  # camera index out of range
    with pytest.raises(ValueError, match="out of range"):
        read_bal(bad)


def test_bal_analytic_blocks_match_finite_differences():
    p = read_bal(TINY)
    Jc, Jp = o.bal_jacobian_blocks(p.cams, p.pts, p.cam_idx, p.pt_idx)
    h = 1e-6
    for k in range(9):
        d = np.zeros_like(p.cams); d[:, k] = h
        fd = (o.bal_residuals(p.cams + d, p.pts, p.cam_idx, p.pt_idx, p.uv)
              - o.bal_residuals(p.cams - d, p.pts, p.cam_idx, p.pt_idx, p.uv)) / (2 * h)
        assert np.abs(fd - Jc[:, :, k]).max() <= 1e-6 * max(1.0, np.abs(Jc[:, :, k]).max()), k
    for k in range(3):
        d = np.zeros_like(p.pts); d[:, k] = h
        fd = (o.bal_residuals(p.cams, p.pts + d, p.cam_idx, p.pt_idx, p.uv)
              - o.bal_residuals(p.cams, p.pts - d, p.cam_idx, p.pt_idx, p.uv)) / (2 * h)
        assert np.abs(fd - Jp[:, :, k]).max() <= 1e-6 * max(1.0, np.abs(Jp[:, :, k]).max()), k


def test_pinhole_conversion_keeps_every_residual():
    p = read_bal(TINY)
    cams = p.cams.copy()
    cams[:, 6], cams[:, 7:] = 820.0, 0.0                                    # one focal length, no distortion
    q = BALProblem(cams, p.pts, p.cam_idx, p.pt_idx, p.uv)
    pin = to_pinhole(q)
    r_bal = o.bal_residuals(q.cams, q.pts, q.cam_idx, q.pt_idx, q.uv)
    r_pin = o.residuals(pin.cams, pin.pts, pin.cam_idx, pin.pt_idx, pin.uv, pin.K4)
    assert np.abs(r_bal - r_pin).max() <= 1e-8          # (one rotation matrix -> vector -> matrix round trip)
    back = from_pinhole(pin)
    assert np.abs(o.bal_residuals(back.cams, back.pts, back.cam_idx, back.pt_idx, back.uv) - r_bal).max() <= 1e-8
    with pytest.raises(ValueError):
        to_pinhole(p)                                                       # distinct f, non-zero k1 / k2


@pytest.mark.gpu
def test_device_bal_residuals_match_the_oracle():
    from bundle_adjustment_amd import hip_backend
    p = read_bal(TINY)
    with hip_backend.Solver(0) as s:
        r, sse, cost = s.residuals_bal(p, "huber")
        ref = o.bal_residuals(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv)
        assert np.abs(r - ref).max() <= 1e-9
        assert abs(sse - float((ref * ref).sum())) <= 1e-10 * sse
        assert abs(cost - o.robust_cost(ref, "huber")) <= 1e-10 * cost
        # a larger synthetic problem in the BAL convention
        rng = np.random.default_rng(5)
        big = BALProblem(np.tile(p.cams, (40, 1)) + rng.normal(0, 1e-3, (240, 9)), np.tile(p.pts, (50, 1)) + rng.normal(0, 0.01, (2000, 3)),
                         rng.integers(0, 240, 30000).astype(np.int32), rng.integers(0, 2000, 30000).astype(np.int32),
                         rng.normal(0, 100, (30000, 2)))
        r, sse, _ = s.residuals_bal(big, "linear")
        assert np.abs(r - o.bal_residuals(big.cams, big.pts, big.cam_idx, big.pt_idx, big.uv)).max() <= 1e-9 * max(1.0, np.abs(r).max())


@pytest.mark.gpu
def test_undistorted_bal_problem_is_adjusted_through_the_reference_model():
    """A BAL file whose cameras share f and have k1 = k2 = 0 converts to the reference's pinhole (negative fy for BAL's
    axis convention) and goes through the LM / Schur / PCG solver; the BAL residuals of the result drop to the noise."""
    from bundle_adjustment_amd import hip_backend
    p = read_bal(TINY)
    cams = p.cams.copy()
    cams[:, 6], cams[:, 7:] = 820.0, 0.0
    truth = BALProblem(cams, p.pts, p.cam_idx, p.pt_idx, np.zeros_like(p.uv))
    uv = -o.bal_residuals(truth.cams, truth.pts, truth.cam_idx, truth.pt_idx, truth.uv) + np.random.default_rng(1).normal(0, 0.3, p.uv.shape)
    rng = np.random.default_rng(2)
    start = BALProblem(cams + np.concatenate([rng.normal(0, 2e-3, (6, 6)), np.zeros((6, 3))], axis=1) * (np.arange(6)[:, None] > 0),
                       p.pts + rng.normal(0, 0.03, p.pts.shape), p.cam_idx, p.pt_idx, uv)
    pin = to_pinhole(start)
    with hip_backend.Solver(0) as s:
        s.set_problem(pin)
        out = s.solve(loss="linear", max_iters=40, ftol=1e-12, xtol=1e-12, gtol=0.0)
        c6, pts = s.get_params()
        assert out["final_sse"] < 0.05 * out["initial_sse"]
        done = from_pinhole(type(pin)(c6, pts, pin.cam_idx, pin.pt_idx, pin.uv, pin.K4, 0))
        r = o.bal_residuals(done.cams, done.pts, done.cam_idx, done.pt_idx, done.uv)
        assert abs(float((r * r).sum()) - out["final_sse"]) <= 1e-8 * out["final_sse"]
        assert np.sqrt((r * r).sum() / start.n_obs) < 0.45
