"""Diagnostic (GPU box): bench batch (drone M = 1e5, S = 50): (a) per-cut trace of the subproblem where the CVaR rows
switch on, (b) the 60-iteration SCP at several cut tolerances: wall-clock, cuts, last L2 change."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_gpu_scp import _bench_batch                      # noqa: E402
from riskaversetrajopt_amd import scp                             # noqa: E402

d, _ = _bench_batch("drone")
us = d.initial_guess_us_mat()
for k in range(2):
    us, _, _ = d.solve_reduced(us, k)
if "trace" in sys.argv:
    d.solve_reduced(us, 2, tol=1e-10, verbose=True)
for tol in (1e-8, 3e-9, 1e-9):
    for rep in range(2):
        d._cut_solver = None
        orig = d.solve_reduced
        d.solve_reduced = lambda u, k, _o=orig, _t=tol: _o(u, k, tol=_t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = scp.run_drone_reduced(d, num_scp_iters_max=60)
        wall = time.perf_counter() - t0
        d.solve_reduced = orig
    print(f"tol {tol:.0e}: cumulative {out['cumulative_s'][-1]:.4f}s wall {wall:.4f}s cuts total {int(out['cuts'].sum())} max {int(out['cuts'].max())} "
          f"first10 {out['cuts'][:10].tolist()} L2 last {out['L2_error'][-1]:.2e}", flush=True)
    np.save(f"gpurun_out/us_tol_{tol:.0e}.npy", out["us"])
a, b, c = (np.load(f"gpurun_out/us_tol_{t:.0e}.npy") for t in (1e-8, 3e-9, 1e-9))
print("final iterate: |u(1e-8) - u(1e-9)| %.2e  |u(3e-9) - u(1e-9)| %.2e" % (np.abs(a - c).max(), np.abs(b - c).max()))
