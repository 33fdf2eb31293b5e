"""GPU: the launch structure of the LM step is pure scheduling.  ba_solve folds three pieces of work into launches that are
running anyway ("riders", csrc/ba_kernels.hpp): the camera update into the back substitution (BA_RIDERS bit 0), the step's
scalar fold + verdict into the speculated point half (bit 1), and -- round 4 -- the back substitution itself into the PCG
point pass whose probe finds PCG finished (bit 2).  Every combination has to produce the SAME BITS as launches of their
own (BA_RIDERS=0): parameters, costs, iteration counts, per-iteration records.  The path this replaces in the reference is
one call, /root/reference/src/bundle_adjuster.py:170-174."""
import numpy as np
import pytest

from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.synthetic import make_problem

pytestmark = pytest.mark.gpu


def _solve(p, monkeypatch, riders, **kw):
    monkeypatch.setenv("BA_RIDERS", str(riders))
    with hip_backend.Solver(0) as s:
        s.set_problem(p)
        out = s.solve(**kw)
        cams, pts = s.get_params()
        trace = s.trace()
        out["cap_floor_raises"] = s.stats()["cap_floor_raises"]
        out["kernels"] = s.profile(reset=True)
    return out, cams, pts, trace


CASES = {
    # every window in LDS, one point-pass workgroup per compute unit or fewer: all three riders apply
    "mid": (dict(n_cams=300, n_pts=30000, obs_per_pt=6, seed=3, outlier_frac=0.02), dict(loss="huber", max_iters=8, ftol=0.0, xtol=0.0, gtol=0.0)),
    # a PCG budget the inner solves run into: the fused probe never sees "finished", the back substitution is a launch of its own
    "capped": (dict(n_cams=120, n_pts=9000, obs_per_pt=5, seed=4), dict(loss="huber", max_iters=6, ftol=0.0, xtol=0.0, gtol=0.0, pcg_max_iters=2)),
    # a start far enough away that steps get rejected and the same linearisation is damped again
    "rejections": (dict(n_cams=60, n_pts=4000, obs_per_pt=5, seed=5, rot_sigma=0.05, trans_sigma=0.3, point_sigma=0.5),
                   dict(loss="linear", max_iters=12, ftol=0.0, xtol=0.0, gtol=0.0, initial_lambda=1e-6)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_every_rider_combination_gives_the_bits_of_separate_launches(case, monkeypatch):
    pk, sk = CASES[case]
    p = make_problem(**pk)
    ref = _solve(p, monkeypatch, 0, **sk)
    if case == "rejections":
        assert any(not r["accepted"] for r in ref[3]), "the case is meant to contain rejected steps"
    if case == "capped":
        assert ref[0]["cap_floor_raises"] > 0, "the case is meant to run into the PCG budget"
    sk = dict(sk, profile=1)
    for riders in (1, 3, 7):
        out, cams, pts, trace = _solve(p, monkeypatch, riders, **sk)
        fused = out["kernels"].get("schur_pt_then_backsub", {}).get("launches", 0)
        if case == "mid":           # the fused probe really ran (inner solves of up to 16 iterations), and never without bit 2
            assert (fused >= out["iterations"] // 2) if riders == 7 else (fused == 0), (riders, fused)
        if case == "capped":         # the solves that ran into the budget ended with a back substitution of its own
            assert fused < out["iterations"]
        assert out["final_cost"] == ref[0]["final_cost"] and out["iterations"] == ref[0]["iterations"], riders
        assert out["pcg_iterations"] == ref[0]["pcg_iterations"] and out["accepted"] == ref[0]["accepted"], riders
        assert np.array_equal(cams, ref[1]) and np.array_equal(pts, ref[2]), riders
        for a, b in zip(trace, ref[3]):
            assert (a["cost_trial"], a["gain_ratio"], a["damping"], a["pcg_iterations"]) == \
                   (b["cost_trial"], b["gain_ratio"], b["damping"], b["pcg_iterations"]), riders
