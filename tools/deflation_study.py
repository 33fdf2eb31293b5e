"""CPU study (oracle operator, numpy): how many PCG iterations a chain-topology problem needs with
Schur-Jacobi alone and with deflation of piecewise-constant camera-segment modes on top of it.
Not part of the product; evidence for DESIGN.md section 8 (what a stronger preconditioner would buy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd.synthetic import make_bal_like
from oracle import ba_oracle as o


def deflated_pcg(op, rhs, Minv, Wd, tol, max_iters):
    """Deflated PCG (Saad, Yeung, Erhel, Guyomarc'h 2000): search directions kept S-orthogonal to span(Wd)."""
    nc = rhs.shape[0]
    n = 6 * nc
    A = lambda v: op.apply(v.reshape(nc, 6)).ravel()
    Mi = lambda v: np.einsum('cij,cj->ci', Minv, v.reshape(nc, 6)).ravel()
    AW = np.stack([A(Wd[:, j]) for j in range(Wd.shape[1])], axis=1)
    E = Wd.T @ AW
    Ei = np.linalg.inv(E)
    b = rhs.ravel()
    x = Wd @ (Ei @ (Wd.T @ b))                 # x0 with W^T r0 = 0
    r = b - A(x)
    z = Mi(r)
    p = z - Wd @ (Ei @ (AW.T @ z))
    rz = r @ z
    rz0 = rhs.ravel() @ Mi(rhs.ravel())
    it = 0
    while it < max_iters:
        q = A(p)
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = Mi(r)
        rz_new = r @ z
        it += 1
        if rz_new <= tol * tol * rz0:
            break
        beta = rz_new / rz
        rz = rz_new
        p = z + beta * p - Wd @ (Ei @ (AW.T @ z))
    return x.reshape(nc, 6), it


def main():
    n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    scale = n_cams / 1723.0
    p = make_bal_like(n_cams=n_cams, n_pts=int(156502 * scale), n_obs_target=int(678718 * scale), seed=0)
    print(f"{p.n_cams} cams / {p.n_pts} pts / {p.n_obs} obs")
    # a few LM iterations first so that lambda and the weights are those of the expensive late iterations
    out = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber", max_iters=6, ftol=0, xtol=0, gtol=0,
                     pcg_tol=0.1, pcg_max_iters=400)
    cams, pts = out["cams"], out["pts"]
    ne = o.normal_equations(cams, pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber")
    for lam in (1e-4, 1e-6):
        op = o.SchurOperator(ne, p.cam_idx, p.pt_idx, lam, 0)
        Minv = np.linalg.inv(op.Hccd - op.schur_diag_blocks())
        Minv[0] = np.eye(6)
        rhs = op.rhs()
        _, it0, _ = o.pcg(op, rhs, Minv, 0.1, 2000)
        _, it0b, _ = o.pcg(op, rhs, Minv, 0.01, 4000)
        print(f"lambda {lam:g}: Schur-Jacobi PCG iterations to 0.1: {it0}, to 0.01: {it0b}")
        for nseg in (4, 8, 16, 32):
            seg = (np.arange(p.n_cams) * nseg) // p.n_cams
            cols = []
            for s_ in range(nseg):
                for d in range(6):
                    v = np.zeros((p.n_cams, 6))
                    v[seg == s_, d] = 1.0
                    v[0] = 0.0
                    if np.abs(v).sum() > 0:
                        cols.append(v.ravel())
            Wd = np.stack(cols, axis=1)
            _, it1 = deflated_pcg(op, rhs, Minv, Wd, 0.1, 2000)
            _, it2 = deflated_pcg(op, rhs, Minv, Wd, 0.01, 4000)
            print(f"   + deflation, {nseg} segments x 6 = {Wd.shape[1]} vectors: {it1} (to 0.1), {it2} (to 0.01)")


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "cluster"):
    main()


def cluster_jacobi_study(n_cams=400):
    """Same problem: PCG iterations with cluster-Jacobi preconditioners (diagonal blocks of m consecutive
    cameras of the explicit Schur complement) instead of the per-camera Schur-Jacobi blocks."""
    scale = n_cams / 1723.0
    p = make_bal_like(n_cams=n_cams, n_pts=int(156502 * scale), n_obs_target=int(678718 * scale), seed=0)
    out = o.lm_solve(p.cams, p.pts, p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber", max_iters=6, ftol=0, xtol=0, gtol=0,
                     pcg_tol=0.1, pcg_max_iters=400)
    ne = o.normal_equations(out["cams"], out["pts"], p.cam_idx, p.pt_idx, p.uv, p.K4, 0, "huber")
    for lam in (1e-4, 1e-6):
        op = o.SchurOperator(ne, p.cam_idx, p.pt_idx, lam, 0)
        S = np.stack([op.apply(np.eye(6 * p.n_cams)[j].reshape(p.n_cams, 6)).ravel() for j in range(6 * p.n_cams)], axis=1)
        rhs = op.rhs().ravel()

        def pcg_dense(Mi_apply, tol, max_iters=5000):
            x = np.zeros_like(rhs); r = rhs.copy(); z = Mi_apply(r); pv = z.copy(); rz = r @ z; rz0 = rz; it = 0
            while it < max_iters:
                q = S @ pv; a = rz / (pv @ q); x += a * pv; r -= a * q; z = Mi_apply(r); rzn = r @ z; it += 1
                if rzn <= tol * tol * rz0: break
                pv = z + (rzn / rz) * pv; rz = rzn
            return it
        for m in (1, 2, 4, 8, 16, 32):
            blocks = []
            for c0 in range(0, p.n_cams, m):
                sl = slice(6 * c0, 6 * min(p.n_cams, c0 + m))
                blocks.append((sl, np.linalg.inv(S[sl, sl])))

            def Mi(v):
                out_ = np.empty_like(v)
                for sl, B in blocks:
                    out_[sl] = B @ v[sl]
                return out_
            print(f"lambda {lam:g}: cluster-Jacobi, {m:2d} cameras per block: {pcg_dense(Mi, 0.1)} iterations to 0.1, "
                  f"{pcg_dense(Mi, 0.01)} to 0.01")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "cluster":
    cluster_jacobi_study(int(sys.argv[1]))
