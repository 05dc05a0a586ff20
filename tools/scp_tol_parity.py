"""Per-subproblem distance between the device path and the all-fp64 streaming leg (tests/test_gpu_scp.py:
test_reduced_subproblems_at_the_benchmarked_size) as a function of the cutting-plane loop's stopping violation.
The fp64 leg's iterates are computed once; the device path then follows them once per tolerance (its kept cuts evolve
as in the test)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_scp import _bench_batch
from tests._host_cuts import DroneStreamingOracle, DrivingStreamingOracle
tols = [1e-9, 1e-8, 1e-7, 1e-6, 1e-5]
for system in sys.argv[1:] or ["drone", "driving"]:
    d, o = _bench_batch(system)
    h = (DroneStreamingOracle if system == "drone" else DrivingStreamingOracle)(o)
    iters = 9 if system == "drone" else 7
    us = h.initial_guess_us_mat()
    seq = []
    for k in range(iters):
        uh, th, ih = h.solve_reduced(us, k)
        seq.append((us, uh, th, ih["cuts"]))
        us = uh
    for tol in tols:
        if getattr(d, "_cut_solver", None) is not None:
            d._cut_solver.keep, d._cut_solver.idle = [], {}
        du, dt, cuts = [], [], []
        for k, (us, uh, th, ch) in enumerate(seq):
            ud, td, idv = d.solve_reduced(us, k, tol=tol)
            du.append(np.abs(ud - uh).max()); dt.append(abs(td - th)); cuts.append(idv["cuts"])
        print(system, "tol %.0e" % tol, "du", " ".join("%.1e" % v for v in du), "| dt", " ".join("%.1e" % v for v in dt), "| cuts", cuts,
              "fp64", [s[3] for s in seq], flush=True)
