"""CPU study (numpy / scipy.sparse, explicit reduced camera matrix of a chain problem): PCG iteration counts of candidate
preconditioners for BASELINE config 5's topology at the dampings the late LM iterations run at.  Not part of the
product; evidence for DESIGN.md (which coarse space to build on the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from bundle_adjustment_amd.synthetic import make_bal_like
from oracle import ba_oracle as o


def explicit_S(ne, cam_idx, pt_idx, lam, fixed):
    nc, npt = ne['Hcc'].shape[0], ne['Hpp'].shape[0]
    Hccd = o.damp_blocks(ne['Hcc'], lam)
    Hinv = np.linalg.inv(o.damp_blocks(ne['Hpp'], lam))
    nobs = len(cam_idx)
    rows = (6 * cam_idx[:, None, None] + np.arange(6)[None, :, None] + 0 * np.arange(3)[None, None, :]).ravel()
    cols = (3 * pt_idx[:, None, None] + 0 * np.arange(6)[None, :, None] + np.arange(3)[None, None, :]).ravel()
    W = sp.csr_matrix((ne['W'].ravel(), (rows, cols)), shape=(6 * nc, 3 * npt))
    Hi = sp.block_diag([h for h in Hinv], format='csr') if npt < 2000 else None
    if Hi is None:
        r = (3 * np.arange(npt)[:, None, None] + np.arange(3)[None, :, None] + 0 * np.arange(3)[None, None, :]).ravel()
        c = (3 * np.arange(npt)[:, None, None] + 0 * np.arange(3)[None, :, None] + np.arange(3)[None, None, :]).ravel()
        Hi = sp.csr_matrix((Hinv.ravel(), (r, c)), shape=(3 * npt, 3 * npt))
    S = -(W @ Hi @ W.T).toarray()
    for c in range(nc):
        S[6 * c:6 * c + 6, 6 * c:6 * c + 6] += Hccd[c]
    y0 = np.einsum('pij,pj->pi', Hinv, ne['bp'])
    g = -(ne['bc'].ravel() - W @ y0.ravel())
    sl = slice(6 * fixed, 6 * fixed + 6)
    S[sl, :] = 0; S[:, sl] = 0; S[sl, sl] = np.eye(6); g[sl] = 0
    return S, g


def pcg(S, b, Mi, tol, maxit):
    x = np.zeros_like(b); r = b.copy(); z = Mi(r); p = z.copy(); rz = r @ z; rz0 = rz; it = 0
    while it < maxit:
        q = S @ p
        a = rz / (p @ q)
        x += a * p; r -= a * q
        z = Mi(r); rzn = r @ z; it += 1
        if rzn <= tol * tol * rz0: break
        p = z + (rzn / rz) * p; rz = rzn
    return x, it


def block_jacobi(S, nb=6):
    n = S.shape[0] // nb
    inv = np.stack([np.linalg.inv(S[nb * c:nb * c + nb, nb * c:nb * c + nb]) for c in range(n)])
    return lambda r: np.einsum('cij,cj->ci', inv, r.reshape(n, nb)).ravel()


def agg_P(nc, m, fixed, kind='const'):
    """prolongation over aggregates of m cameras; const: piecewise constant; lin: hat functions with nodes every m cameras"""
    if kind == 'const':
        na = (nc + m - 1) // m
        P = np.zeros((6 * nc, 6 * na))
        for c in range(nc):
            if c == fixed: continue
            for d in range(6): P[6 * c + d, 6 * (c // m) + d] = 1.0
    else:
        nodes = np.arange(0, nc + m, m)
        na = len(nodes)
        P = np.zeros((6 * nc, 6 * na))
        for c in range(nc):
            if c == fixed: continue
            a = c // m; t = (c - nodes[a]) / m
            for d in range(6):
                P[6 * c + d, 6 * a + d] = 1 - t
                if a + 1 < na: P[6 * c + d, 6 * (a + 1) + d] = t
    keep = np.abs(P).sum(0) > 0
    return P[:, keep]


def two_level(S, Mj, P):
    E = P.T @ S @ P
    Ei = np.linalg.inv(E)
    return lambda r: Mj(r) + P @ (Ei @ (P.T @ r))


def multilevel(S, Mj, nc, fixed, sizes, coarse_exact=True, kind='const'):
    """additive multilevel: block-Jacobi on the fine level + for every aggregate size the block-diagonal (6x6 per aggregate)
    inverse of P^T S P, the coarsest level solved exactly"""
    terms = []
    for i, m in enumerate(sizes):
        P = agg_P(nc, m, fixed, kind)
        E = P.T @ S @ P
        if i == len(sizes) - 1 and coarse_exact:
            Ei = np.linalg.inv(E)
        else:
            nb = 6
            Ei = np.zeros_like(E)
            for a in range(E.shape[0] // nb):
                Ei[nb * a:nb * a + nb, nb * a:nb * a + nb] = np.linalg.inv(E[nb * a:nb * a + nb, nb * a:nb * a + nb])
        terms.append((P, Ei))
    def apply(r):
        z = Mj(r)
        for P, Ei in terms: z = z + P @ (Ei @ (P.T @ r))
        return z
    return apply


def schwarz(S, nc, size, overlap, fixed):
    blocks = []
    start = 0
    while start < nc:
        lo, hi = max(0, start - overlap), min(nc, start + size + overlap)
        idx = np.arange(6 * lo, 6 * hi)
        blocks.append((idx, np.linalg.inv(S[np.ix_(idx, idx)])))
        start += size
    def apply(r):
        z = np.zeros_like(r)
        for idx, inv in blocks: z[idx] += inv @ r[idx]
        return z
    return apply


def main():
    n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    scale = n_cams / 1723.0
    p = make_bal_like(n_cams=n_cams, n_pts=int(156502 * scale), n_obs_target=int(678718 * scale), seed=0)
    print(f"{p.n_cams} cams / {p.n_pts} pts / {p.n_obs} obs", flush=True)
    out = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber", max_iters=6, ftol=0, xtol=0, gtol=0, pcg_tol=0.1, pcg_max_iters=400)
    ne = o.normal_equations(out["cams"], out["pts"], p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber")
    nc = p.n_cams
    for lam in (1e-4, 1e-6, 1e-8, 1e-10):
        t = time.time()
        S, g = explicit_S(ne, p.cam_idx, p.pt_idx, lam, 0)
        Mj = block_jacobi(S)
        res = {}
        res["schur-jacobi"] = pcg(S, g, Mj, 0.1, 5000)[1]
        for m in (16, 8, 4):
            res[f"2lvl const m={m}"] = pcg(S, g, two_level(S, Mj, agg_P(nc, m, 0)), 0.1, 5000)[1]
        for m in (16, 8):
            res[f"2lvl linear m={m}"] = pcg(S, g, two_level(S, Mj, agg_P(nc, m, 0, 'lin')), 0.1, 5000)[1]
        res["multilevel const 2,4,8,16"] = pcg(S, g, multilevel(S, Mj, nc, 0, (2, 4, 8, 16)), 0.1, 5000)[1]
        res["multilevel const 4,16"] = pcg(S, g, multilevel(S, Mj, nc, 0, (4, 16)), 0.1, 5000)[1]
        res["schwarz 16+4"] = pcg(S, g, schwarz(S, nc, 16, 4, 0), 0.1, 5000)[1]
        sw = schwarz(S, nc, 16, 4, 0); P16 = agg_P(nc, 16, 0); Ei = np.linalg.inv(P16.T @ S @ P16)
        res["schwarz 16+4 + coarse const 16"] = pcg(S, g, lambda r: sw(r) + P16 @ (Ei @ (P16.T @ r)), 0.1, 5000)[1]
        Pl = agg_P(nc, 16, 0, 'lin'); Eil = np.linalg.inv(Pl.T @ S @ Pl)
        res["schwarz 16+4 + coarse linear 16"] = pcg(S, g, lambda r: sw(r) + Pl @ (Eil @ (Pl.T @ r)), 0.1, 5000)[1]
        print(f"lambda {lam:g} ({time.time() - t:.0f} s): " + "; ".join(f"{k}: {v}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
