import sys; sys.path.insert(0,'.')
from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.synthetic import make_config
p=make_config("C3")
s=hip_backend.Solver(0); s.set_problem(p)
out=s.solve(loss="huber",max_iters=20,ftol=0,xtol=0,gtol=0,verbose=1)
print(out)
