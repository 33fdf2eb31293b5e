"""ctypes binding of ``libba_hip.so`` (C ABI: ``include/ba_hip.h``).

This is the only way the package computes anything: there is no CPU fallback.  If the
shared library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)
or no GPU is visible, the calls raise ``BAHipError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .problem import BAProblem

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BA_HIP_LIB") or os.path.join(_HERE, "libba_hip.so")   # BA_HIP_LIB: another build of the same ABI

LOSS = {"linear": 0, "huber": 1}
PRECOND = {"jacobi": 0, "schur_jacobi": 1, "two_level": 2}
STATUS_NAMES = {0: "max_iters", 1: "ftol", 2: "xtol", 3: "gtol"}
PROFILE_SLOTS = 16
K_RESIDUAL, K_LINEARIZE_CAM, K_LINEARIZE_PT, K_POINT_INVERT, K_SCHUR_PT, K_SCHUR_CAM = 1, 2, 3, 4, 5, 6


class BAHipError(RuntimeError):
    pass


class BAOptions(C.Structure):
    _fields_ = [("loss", C.c_int32), ("max_iters", C.c_int32), ("f_scale", C.c_double), ("ftol", C.c_double),
                ("xtol", C.c_double), ("gtol", C.c_double), ("initial_lambda", C.c_double), ("pcg_tol", C.c_double),
                ("pcg_max_iters", C.c_int32), ("pcg_min_iters", C.c_int32), ("preconditioner", C.c_int32),
                ("jacobian_precision", C.c_int32), ("reserved0", C.c_int32), ("profile", C.c_int32),
                ("verbose", C.c_int32), ("small_solver", C.c_int32), ("pcg_model_tol", C.c_double),
                ("pcg_model_min_iters", C.c_int32), ("precond_lag", C.c_int32)]


class BASummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("accepted", C.c_int32), ("pcg_iterations", C.c_int32),
                ("status", C.c_int32), ("initial_sse", C.c_double), ("final_sse", C.c_double),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_lambda", C.c_double),
                ("seconds_total", C.c_double), ("seconds_linearize", C.c_double), ("seconds_pcg", C.c_double),
                ("seconds_update", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["status_name"] = STATUS_NAMES.get(self.status, str(self.status))
        return d


class BAIterRecord(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("accepted", C.c_int32), ("pcg_iterations", C.c_int32), ("reserved", C.c_int32),
                ("cost", C.c_double), ("cost_trial", C.c_double), ("sse_trial", C.c_double), ("lambda_", C.c_double),
                ("gain_ratio", C.c_double), ("step_norm", C.c_double), ("seconds", C.c_double)]


class BAProfile(C.Structure):
    _fields_ = [("launches", C.c_int32 * PROFILE_SLOTS), ("total_ms", C.c_double * PROFILE_SLOTS),
                ("working_launches", C.c_int32 * PROFILE_SLOTS), ("working_ms", C.c_double * PROFILE_SLOTS)]


_lib = None

# name -> (restype, argtypes); every symbol include/ba_hip.h declares
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)
SYMBOLS = {
    "ba_last_error": (C.c_char_p, []),
    "ba_kernel_name": (C.c_char_p, [C.c_int]),
    "ba_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ba_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ba_destroy": (C.c_int, [C.c_void_p]),
    "ba_synchronize": (C.c_int, [C.c_void_p]),
    "ba_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ba_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ba_set_problem": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, _IP, _IP, _DP, _DP, C.c_int32]),
    "ba_set_params": (C.c_int, [C.c_void_p, _DP, _DP]),
    "ba_get_params": (C.c_int, [C.c_void_p, _DP, _DP]),
    "ba_get_rotations": (C.c_int, [C.c_void_p, _DP]),
    "ba_allgather_points": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _DP]),
    "ba_residuals": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, _DP, _DP, _DP]),
    "ba_residuals_bal": (C.c_int, [C.c_void_p, _DP, C.c_int32, C.c_double, _DP, _DP, _DP]),
    "ba_linearize_bal": (C.c_int, [C.c_void_p, _DP, C.c_int32, C.c_double, _DP, _DP, _DP, _DP]),
    "ba_solve_bal": (C.c_int, [C.c_void_p, _DP, C.POINTER(BAOptions), C.POINTER(BASummary)]),
    "ba_linearize": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, _DP, _DP, _DP, _DP]),
    "ba_schur_rhs": (C.c_int, [C.c_void_p, C.c_double, _DP]),
    "ba_schur_apply": (C.c_int, [C.c_void_p, C.c_double, _DP, _DP]),
    "ba_default_options": (C.c_int, [C.POINTER(BAOptions)]),
    "ba_solve": (C.c_int, [C.c_void_p, C.POINTER(BAOptions), C.POINTER(BASummary)]),
    "ba_get_profile": (C.c_int, [C.c_void_p, C.POINTER(BAProfile)]),
    "ba_reset_profile": (C.c_int, [C.c_void_p]),
    "ba_time_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _DP]),
    "ba_triangulate": (C.c_int, [C.c_void_p, _DP, _DP, _DP, C.c_int64, _DP, _DP, _DP, C.POINTER(C.c_uint8)]),
    "ba_get_trace": (C.c_int, [C.c_void_p, C.POINTER(BAIterRecord), C.c_int32, C.POINTER(C.c_int32)]),
    "ba_get_stat": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "ba_debug_occupy": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double]),
    "ba_debug_layout": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
}
# enum ba_stat (include/ba_hip.h)
STATS = {"window_mw_launches": 0, "window_lm_launches": 1, "window_fallbacks": 2, "precond_builds": 3, "precond_reuses": 4, "banded": 5,
         "cap_floor_raises": 6, "ipc_exchanges": 7, "pixels_f32": 8}


def load_library():
    """Load libba_hip.so and declare every prototype.  Loud failure when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BAHipError(f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                         "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise BAHipError(f"libba_hip error {rc}: {load_library().ba_last_error().decode()}")


def _dp(a):
    return a.ctypes.data_as(_DP) if a is not None else None


def device_count():
    n = C.c_int(0)
    _check(load_library().ba_device_count(C.byref(n)))
    return n.value


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(load_library().ba_comm_unique_id(buf))
    return buf.raw


class Solver:
    """One GPU, one problem.  Thin object wrapper over the handle API."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.ba_create(int(device_id), C.byref(self._h)))
        self.n_cams = self.n_pts = self.n_obs = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- multi-rank ------------------------------------------------------------------
    def comm_init(self, rank, world, unique_id: bytes | None):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        _check(self._lib.ba_comm_init(self._h, int(rank), int(world), buf))

    # -- problem / parameters --------------------------------------------------------
    def set_problem(self, prob: BAProblem, with_params=True):
        prob.validate()
        cam_idx = np.ascontiguousarray(prob.cam_idx, dtype=np.int32)
        pt_idx = np.ascontiguousarray(prob.pt_idx, dtype=np.int32)
        uv = np.ascontiguousarray(prob.uv, dtype=np.float64)
        K4 = np.ascontiguousarray(prob.K4, dtype=np.float64)
        _check(self._lib.ba_set_problem(self._h, prob.n_cams, prob.n_pts, prob.n_obs,
                                        cam_idx.ctypes.data_as(_IP), pt_idx.ctypes.data_as(_IP), _dp(uv), _dp(K4),
                                        int(prob.fixed_cam)))
        self.n_cams, self.n_pts, self.n_obs = prob.n_cams, prob.n_pts, prob.n_obs
        if with_params:
            self.set_params(prob.cams, prob.pts)

    def set_params(self, cams, pts):
        cams = np.ascontiguousarray(cams, dtype=np.float64).reshape(self.n_cams, 6)
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(self.n_pts, 3)
        _check(self._lib.ba_set_params(self._h, _dp(cams), _dp(pts)))

    def get_params(self):
        cams = np.empty((self.n_cams, 6))
        pts = np.empty((self.n_pts, 3))
        _check(self._lib.ba_get_params(self._h, _dp(cams), _dp(pts)))
        return cams, pts

    def allgather_points(self, p_begin, n_total):
        """Multi-rank: the points of every shard, on every rank (collective)."""
        pts = np.empty((int(n_total), 3))
        _check(self._lib.ba_allgather_points(self._h, int(p_begin), int(n_total), _dp(pts)))
        return pts

    def get_rotations(self):
        R = np.empty((self.n_cams, 3, 3))
        _check(self._lib.ba_get_rotations(self._h, _dp(R)))
        return R

    # -- kernels behind the parity entry points ----------------------------------------
    def residuals(self, loss="linear", f_scale=1.0, want_vector=True):
        r = np.empty((self.n_obs, 2)) if want_vector else None
        sse, cost = C.c_double(), C.c_double()
        _check(self._lib.ba_residuals(self._h, LOSS[loss], float(f_scale), _dp(r), C.byref(sse), C.byref(cost)))
        return r, sse.value, cost.value

    def linearize(self, loss="linear", f_scale=1.0):
        Hcc = np.empty((self.n_cams, 21)); bc = np.empty((self.n_cams, 6))
        Hpp = np.empty((self.n_pts, 6)); bp = np.empty((self.n_pts, 3))
        _check(self._lib.ba_linearize(self._h, LOSS[loss], float(f_scale), _dp(Hcc), _dp(bc), _dp(Hpp), _dp(bp)))
        return Hcc, bc, Hpp, bp

    def schur_rhs(self, lam):
        g = np.empty((self.n_cams, 6))
        _check(self._lib.ba_schur_rhs(self._h, float(lam), _dp(g)))
        return g

    def schur_apply(self, lam, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.n_cams, 6)
        out = np.empty((self.n_cams, 6))
        _check(self._lib.ba_schur_apply(self._h, float(lam), _dp(v), _dp(out)))
        return out

    # -- solve -------------------------------------------------------------------------
    def default_options(self) -> BAOptions:
        o = BAOptions()
        _check(self._lib.ba_default_options(C.byref(o)))
        return o

    def solve(self, **kw):
        """kw: loss ('linear'|'huber'), preconditioner ('jacobi'|'schur_jacobi') or any
        ba_options field.  Returns the summary as a dict."""
        o = self._options(kw)
        s = BASummary()
        _check(self._lib.ba_solve(self._h, C.byref(o), C.byref(s)))
        return s.as_dict()

    def residuals_bal(self, bal, loss="linear", f_scale=1.0, want_vector=True):
        """BAL 9-parameter camera residuals (bal.BALProblem) on the GPU: uploads the problem, returns (r, sse, cost)."""
        from .problem import BAProblem
        self.set_problem(BAProblem(np.ascontiguousarray(bal.cams[:, :6]), bal.pts, bal.cam_idx, bal.pt_idx, bal.uv,
                                   np.array([1.0, 1.0, 0.0, 0.0]), -1))
        intr = np.ascontiguousarray(bal.cams[:, 6:9], dtype=np.float64)
        r = np.empty((self.n_obs, 2)) if want_vector else None
        sse, cost = C.c_double(0), C.c_double(0)
        _check(self._lib.ba_residuals_bal(self._h, _dp(intr), LOSS[loss] if isinstance(loss, str) else loss, float(f_scale),
                                          _dp(r), C.byref(sse), C.byref(cost)))
        return r, sse.value, cost.value

    def _set_bal(self, bal, fixed_cam=-1):
        from .problem import BAProblem
        self.set_problem(BAProblem(np.ascontiguousarray(bal.cams[:, :6]), bal.pts, bal.cam_idx, bal.pt_idx, bal.uv,
                                   np.array([1.0, 1.0, 0.0, 0.0]), fixed_cam))
        return np.ascontiguousarray(bal.cams[:, 6:9], dtype=np.float64).copy()

    def linearize_bal(self, bal, loss="linear", f_scale=1.0, fixed_cam=-1):
        """ba_linearize_bal on a bal.BALProblem: dict(Hcc (Nc,45), bc (Nc,9), Hpp (Np,6), bp (Np,3)), packed upper triangles."""
        intr = self._set_bal(bal, fixed_cam)
        out = dict(Hcc=np.empty((self.n_cams, 45)), bc=np.empty((self.n_cams, 9)), Hpp=np.empty((self.n_pts, 6)),
                   bp=np.empty((self.n_pts, 3)))
        _check(self._lib.ba_linearize_bal(self._h, _dp(intr), LOSS[loss] if isinstance(loss, str) else loss, float(f_scale),
                                          _dp(out["Hcc"]), _dp(out["bc"]), _dp(out["Hpp"]), _dp(out["bp"])))
        return out

    def set_problem_bal(self, bal, fixed_cam=-1):
        """Upload a bal.BALProblem once (observation lists, poses, points); returns its (Nc,3) intrinsics.  Solves on the
        resident problem: ``solve_bal_resident``; parameters are restored with ``set_params(bal.cams[:, :6], bal.pts)``."""
        return self._set_bal(bal, fixed_cam)

    def solve_bal_resident(self, intr, **kw):
        """ba_solve_bal on the problem the handle already holds: intr (Nc,3) = (f, k1, k2) per camera at the start, adjusted
        in place.  Returns the summary dict."""
        intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(self.n_cams, 3)
        o = self._options(kw)
        s = BASummary()
        _check(self._lib.ba_solve_bal(self._h, _dp(intr), C.byref(o), C.byref(s)))
        return s.as_dict()

    def _options(self, kw):
        """ba_default_options overlaid with kw (loss / preconditioner by name or by value, any other ba_options field)."""
        o = self.default_options()
        for k, v in kw.items():
            if k == "loss":
                v = LOSS[v] if isinstance(v, str) else v
            if k == "preconditioner":
                v = PRECOND[v] if isinstance(v, str) else v
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, v)
        return o

    def solve_bal(self, bal, fixed_cam=-1, **kw):
        """ba_solve_bal on a bal.BALProblem (9-parameter cameras, f / k1 / k2 adjusted with the pose): returns
        (summary dict, cams (Nc,9), pts (Np,3)).  kw as for solve()."""
        intr = self._set_bal(bal, fixed_cam)
        o = self._options(kw)
        s = BASummary()
        _check(self._lib.ba_solve_bal(self._h, _dp(intr), C.byref(o), C.byref(s)))
        cams6, pts = self.get_params()
        return s.as_dict(), np.concatenate([cams6, intr], axis=1), pts

    def triangulate(self, camera_matrix, R_rel, t_rel, pts1, pts2):
        """ba_triangulate: (n,3) points in the first camera's frame and the (n,) cheirality mask."""
        K = np.ascontiguousarray(camera_matrix, dtype=np.float64).reshape(3, 3)
        R = np.ascontiguousarray(R_rel, dtype=np.float64).reshape(3, 3)
        t = np.ascontiguousarray(t_rel, dtype=np.float64).reshape(3)
        p1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
        p2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
        n = p1.shape[0]
        xyz = np.empty((n, 3))
        valid = np.zeros(n, dtype=np.uint8)
        _check(self._lib.ba_triangulate(self._h, _dp(K), _dp(R), _dp(t), n, _dp(p1), _dp(p2), _dp(xyz),
                                        valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return xyz, valid.astype(bool)

    def trace(self):
        """Per-iteration records of the last solve (ba_get_trace): list of dicts."""
        n = C.c_int32(0)
        _check(self._lib.ba_get_trace(self._h, None, 0, C.byref(n)))
        if n.value == 0:
            return []
        buf = (BAIterRecord * n.value)()
        _check(self._lib.ba_get_trace(self._h, buf, n.value, C.byref(n)))
        return [dict(iteration=r.iteration, accepted=bool(r.accepted), pcg_iterations=r.pcg_iterations, cost=r.cost,
                     cost_trial=r.cost_trial, sse_trial=r.sse_trial, damping=r.lambda_, gain_ratio=r.gain_ratio,
                     step_norm=r.step_norm, seconds=r.seconds) for r in buf[:n.value]]

    def profile(self, reset=False):
        p = BAProfile()
        _check(self._lib.ba_get_profile(self._h, C.byref(p)))
        out = {}
        for i in range(PROFILE_SLOTS):
            if p.launches[i]:
                w = max(p.working_launches[i], 1)
                out[self._lib.ba_kernel_name(i).decode()] = dict(
                    launches=p.launches[i], total_ms=p.total_ms[i], mean_us=1e3 * p.total_ms[i] / p.launches[i],
                    working_launches=p.working_launches[i], working_mean_us=1e3 * p.working_ms[i] / w)
        if reset:
            _check(self._lib.ba_reset_profile(self._h))
        return out

    def time_kernel(self, slot, reps=50):
        us = C.c_double()
        _check(self._lib.ba_time_kernel(self._h, int(slot), int(reps), C.byref(us)))
        return us.value

    def synchronize(self):
        _check(self._lib.ba_synchronize(self._h))

    def stats(self):
        """Event counters of the handle (ba_get_stat): dict name -> count since the handle was created."""
        out = {}
        for name, which in STATS.items():
            v = C.c_int64(0)
            _check(self._lib.ba_get_stat(self._h, which, C.byref(v)))
            out[name] = v.value
        return out

    LAYOUT = {"pt_off": 0, "p_cam": 1, "c_pt": 2, "c_orig": 3, "offk": 4, "long_pts": 5, "blk_win": 6, "slot": 7, "scalars": 8,
              "p_uv": 9, "c_uv": 10}
    LAYOUT_SCALARS = ("lanes", "nblkP", "ppb", "nblkL", "long_spb", "long_thr", "n_long", "cam_band", "banded", "cam_segl",
                      "all_lds_pinhole", "all_lds_bal", "lds_bytes_pinhole", "lds_bytes_bal", "build_path", "mw_ok")

    def debug_layout(self, name):
        """Test hook (ba_debug_layout): one array of the layout ba_set_problem built, as numpy (scalars: a dict)."""
        which = self.LAYOUT[name]
        cap = max(2 * self.n_obs, 9 * self.n_cams, self.n_pts + 1, 16384)
        buf = np.empty(cap, dtype=np.float64 if which >= 9 else np.int32)
        n = C.c_int64(0)
        _check(self._lib.ba_debug_layout(self._h, which, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        out = buf[:n.value].copy()
        return dict(zip(self.LAYOUT_SCALARS, (int(v) for v in out))) if name == "scalars" else out

    def debug_occupy(self, n_workgroups, lds_bytes, milliseconds):
        """Test hook (ba_debug_occupy): idle workgroups on a second stream hold compute units' LDS for a bounded time."""
        _check(self._lib.ba_debug_occupy(self._h, int(n_workgroups), int(lds_bytes), float(milliseconds)))
