"""Diagnostic (GPU box): the subproblem where the CVaR rows switch on, bench batch (M = 1e5), device path at several cut
tolerances against the streaming fp64 leg at 1e-10 / 1e-12 from the same iterate."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tests.test_gpu_scp import _bench_batch                      # noqa: E402
from tests._host_cuts import DroneStreamingOracle, DrivingStreamingOracle   # noqa: E402

system = sys.argv[1] if len(sys.argv) > 1 else "drone"
d, o = _bench_batch(system)
h = (DroneStreamingOracle if system == "drone" else DrivingStreamingOracle)(o)
first = 2 if system == "drone" else 1
us = h.initial_guess_us_mat()
for k in range(first):
    us, _, _ = h.solve_reduced(us, k)
t0 = time.time()
ref = {}
for tol in (1e-10, 1e-12):
    h.cs.keep, h.cs.idle = [], {}
    uh, th, ih = h.solve_reduced(us, first, tol=tol)
    ref[tol] = (uh, th)
    print(f"host tol {tol:.0e}: cuts {ih['cuts']} slack {ih['slack']:.3e} t_risk {th:.9f}  ({time.time() - t0:.1f}s)", flush=True)
print("host 1e-10 vs 1e-12: |du| %.2e" % np.abs(ref[1e-10][0] - ref[1e-12][0]).max())
for tol in (1e-7, 1e-8, 1e-9, 1e-10, 1e-11):
    d._cut_solver = None
    d.solve_reduced(us, 0)
    ud, td, idv = d.solve_reduced(us, first, tol=tol)
    print(f"device tol {tol:.0e}: cuts {idv['cuts']} status {idv['status']}  |du| vs host(1e-10) %.2e vs host(1e-12) %.2e  |dt| %.2e"
          % (np.abs(ud - ref[1e-10][0]).max(), np.abs(ud - ref[1e-12][0]).max(), abs(td - ref[1e-12][1])), flush=True)
