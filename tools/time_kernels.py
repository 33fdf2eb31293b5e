"""Isolated back-to-back timings of the solver's kernels on the C3 problem (ba_time_kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import hip_backend as hb
from bundle_adjustment_amd.synthetic import make_config
p = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
s = hb.Solver(0); s.set_problem(p)
s.linearize("huber")
for name, slot in (("residual", 1), ("linearize_cam", 2), ("linearize_pt", 3), ("point_invert", 4), ("schur_pt", 5), ("schur_cam", 6), ("rhs+diag", 8)):
    print(f"{name:16s} {s.time_kernel(slot, 200):8.2f} us")
