"""Test double for hip_backend.Solver backed by the CPU oracle.  Lets the host-side logic
of BundleAdjuster (window selection, packing, skip / divergence paths, write-back, log
lines) be tested where there is no GPU.  Test infrastructure only."""
import numpy as np

from oracle import ba_oracle as o


class OracleSolver:
    force_diverge = False

    def __init__(self, device_id=0):
        self.prob = None

    def close(self):
        pass

    def set_problem(self, prob, with_params=True):
        prob.validate()
        self.prob = prob
        self.cams, self.pts = prob.cams.copy(), prob.pts.copy()
        self.n_cams, self.n_pts, self.n_obs = prob.n_cams, prob.n_pts, prob.n_obs

    def set_params(self, cams, pts):
        self.cams, self.pts = np.array(cams, dtype=np.float64), np.array(pts, dtype=np.float64)

    def residuals(self, loss="linear", f_scale=1.0, want_vector=True):
        p = self.prob
        r = o.residuals(self.cams, self.pts, p.cam_idx, p.pt_idx, p.uv, p.K4)
        return r, float((r * r).sum()), o.robust_cost(r, loss)

    def solve(self, **kw):
        p = self.prob
        out = o.lm_solve(self.cams, self.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, p.fixed_cam, kw.get("loss", "huber"),
                         max_iters=kw.get("max_iters", 50), ftol=kw.get("ftol", 1e-5), xtol=kw.get("xtol", 1e-5),
                         gtol=kw.get("gtol", 1e-8), pcg_tol=kw.get("pcg_tol", 0.1))
        if self.force_diverge:
            return dict(initial_sse=out["sse0"], final_sse=out["sse0"] * 1.5, iterations=1, accepted=0)
        self.cams, self.pts = out["cams"], out["pts"]
        return dict(initial_sse=out["sse0"], final_sse=out["sse"], initial_cost=out["cost0"], final_cost=out["cost"],
                    iterations=out["iterations"], accepted=out["accepted"], pcg_iterations=out["pcg_iters"], seconds_total=1e-3)

    def get_params(self):
        return self.cams.copy(), self.pts.copy()

    def get_rotations(self):
        return o.rodrigues_batch(self.cams[:, :3])


class GlooOracleSolver(OracleSolver):
    """Multi-rank double (one process per rank, torch.distributed gloo already initialised): what the
    library's collectives do is restated with gloo -- the shards are gathered, every rank solves the
    reassembled problem with the oracle (identical arithmetic on every rank, like the replicated
    camera side of the real solver), keeps its own landmark block, and allgather_points exchanges
    the blocks."""

    def comm_init(self, rank, world, unique_id):
        self.rank, self.world = rank, world

    def solve(self, **kw):
        import torch.distributed as dist
        from bundle_adjustment_amd.problem import BAProblem
        mine = self.prob
        parts = [None] * self.world
        dist.all_gather_object(parts, (mine.pts, mine.cam_idx, mine.pt_idx, mine.uv))
        offs = np.cumsum([0] + [q[0].shape[0] for q in parts])
        full = BAProblem(mine.cams.copy(), np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts]),
                         np.concatenate([q[2] + offs[i] for i, q in enumerate(parts)]).astype(np.int32),
                         np.concatenate([q[3] for q in parts]), mine.K4, mine.fixed_cam)
        self.prob, self.cams, self.pts = full, full.cams.copy(), full.pts.copy()
        out = super().solve(**kw)
        b, e = offs[self.rank], offs[self.rank + 1]
        self.prob, self.pts = mine, self.pts[b:e].copy()
        return out

    def allgather_points(self, p_begin, n_total):
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, (int(p_begin), self.pts))
        full = np.zeros((int(n_total), 3))
        for b, pts in parts:
            full[b:b + pts.shape[0]] = pts
        return full
