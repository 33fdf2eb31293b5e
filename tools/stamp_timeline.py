#!/usr/bin/env python3
"""Where a PCG kernel's time goes, from in-kernel clock stamps (diagnostic build only).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DBA_STAMPS \
          -o bundle_adjustment_amd/libba_hip_stamps.so bundle_adjustment_amd/csrc/ba_hip.hip -ldl -lrt
    BA_HIP_LIB=bundle_adjustment_amd/libba_hip_stamps.so python tools/stamp_timeline.py [C3]

Thread 0 of every workgroup stamps s_memrealtime (100 MHz) at entry / exit and s_memtime (shader clock) at the
stages marked BA_STAMP in csrc/ba_kernels.hpp.  Printed per kernel: the spread of workgroup start times, the
launch's span (first start -> last exit), and per stage the shader cycles since entry (median, p10, p90, max).
The stamps come from the LAST launch of each kernel in a short solve, so PCG must not have converged in it:
pcg_tol = 0 and pcg_max_iters = 3 make every launch a working one.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend as hb          # noqa: E402
from bundle_adjustment_amd.synthetic import make_bal_like, make_bal_problem, make_config   # noqa: E402

STAGES = {
    0: ("k_pt_schur<MODE 0>", ["entry", "verdict known", "table filled", "main loop done", "points stored (pre block sum)", "exit"], [1, 2, 3, 4, 5, 6]),
    1: ("k_cam_schur<PCG>", ["entry", "verdict known", "main loop done", "exit"], [1, 2, 3, 6]),
    2: ("k_pcg_step", ["entry", "verdict known", "update done", "exit"], [1, 2, 3, 6]),
}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    bal = make_bal_problem(seed=0) if cfg == "C5bal" else None          # config 5 with the BAL 9-parameter camera
    p = bal if bal is not None else (make_bal_like(seed=0) if cfg == "C5" else make_config(cfg, seed=0))
    s = hb.Solver(0)
    lib = s._lib
    lib.ba_debug_stamps.restype = C.c_int
    lib.ba_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    intr = s.set_problem_bal(bal, fixed_cam=0) if bal is not None else s.set_problem(p)
    mode = int(os.environ.get("BA_DBG_MODE", "0"))
    if mode:
        lib.ba_debug_mode.restype = C.c_int
        lib.ba_debug_mode.argtypes = [C.c_void_p, C.c_int]
        hb._check(lib.ba_debug_mode(s._h, mode))
        print(f"camera-pass gather variant {mode} (results are meaningless; timing only)")
    kw = dict(loss="huber", max_iters=3, ftol=0.0, xtol=0.0, gtol=0.0, pcg_tol=0.0, pcg_max_iters=3, pcg_min_iters=3)
    if bal is not None:
        s.solve_bal_resident(intr, **kw)
    else:
        s.solve(**kw)
    vc = 8 if bal is not None else 16
    nblk = {0: min(8192, (p.n_pts + 511) // 512 + 64), 1: min(8192, ((p.n_cams + 3) // 4) * 8), 2: (p.n_cams + vc - 1) // vc}
    for kind, (name, labels, slots) in STAGES.items():
        n = nblk[kind]
        buf = (C.c_uint64 * (n * 8))()
        hb._check(lib.ba_debug_stamps(s._h, kind, buf, n))
        a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
        a = a[(a[:, 0] > 0) & (a[:, 7] >= a[:, 0])]
        if a.shape[0] == 0:
            print(name, ": no stamps")
            continue
        t0 = a[:, 0].min()
        start_us = (a[:, 0] - t0) / 100.0
        end_us = (a[:, 7] - t0) / 100.0
        life = a[:, 6] - a[:, 1]
        clk = np.median(life / np.maximum((a[:, 7] - a[:, 0]) / 100.0, 1e-9)) / 1e3      # shader GHz
        print(f"\n{name}: {a.shape[0]} workgroups stamped; starts spread over {start_us.max():.2f} us "
              f"(median {np.median(start_us):.2f}); span first start -> last exit {end_us.max():.2f} us; "
              f"workgroup life median {np.median(end_us - start_us):.2f} us, max {(end_us - start_us).max():.2f} us; "
              f"shader clock ~{clk:.2f} GHz")
        for lab, sl in zip(labels, slots):
            d = (a[:, sl] - a[:, 1])
            d = d[a[:, sl] > 0]
            if d.size == 0:
                continue
            print(f"  {lab:32s} cycles since entry: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  "
                  f"p90 {np.percentile(d, 90):8.0f}  max {d.max():8.0f}   (= {np.median(d) / (clk * 1e3):6.2f} us)")
    s.close()


if __name__ == "__main__":
    main()
