import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import drone as od
from tests import _host_cuts as hc
from riskaversetrajopt_amd import scp
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
S = 50
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rng = np.random.RandomState(7)
t0=time.time()
DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
om = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
print("sampled", time.time()-t0)
mdl = hc.DroneStreamingOracle(om, nthreads=8)
us = mdl.initial_guess_us_mat()
tot=0
for k in range(iters):
    t0=time.time()
    us_new, t_risk, info = mdl.solve_reduced(us, k, tol=1e-9)
    err = scp.L2_error_us(us_new, us)
    tot += info["cuts"]
    print(f"scp {k:2d} cuts {info['cuts']:3d} recycled {info.get('recycled',0):2d} status {info['status']} L2 {err:.3e}  {time.time()-t0:.2f}s", flush=True)
    us = us_new
print("total cuts", tot)
