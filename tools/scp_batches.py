"""SCP wall-clock / convergence over 12 device-sampled batches (drone M = 1e5, S = 50, alpha = 0.1, 60 iterations, the
reference's timing protocol) at two cut tolerances.   usage: python tools/scp_batches.py [n_seeds]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from riskaversetrajopt_amd import drone_risk, drone_utils, scp   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
M, S = 100000, 50
for tol in (1e-8, 1e-9):
    tot = []
    for seed in range(n):
        dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
        d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
        d.solve_reduced(d.initial_guess_us_mat(), 2, tol=tol)
        d._cut_solver = None
        torch.cuda.synchronize()
        out = scp.run_drone_reduced(d, num_scp_iters_max=60, tol=tol)
        tot.append(out["cumulative_s"][-1])
        print(f"tol {tol:.0e} seed {seed:2d} cumulative {out['cumulative_s'][-1]:.4f} define {np.median(out['define_s']):.2e} "
              f"solve {np.median(out['solve_s']):.2e} cuts {int(out['cuts'].sum()):4d} max {int(out['cuts'].max()):3d} "
              f"L2 last {out['L2_error'][-1]:.1e}", flush=True)
    print(f"tol {tol:.0e}: cumulative min {min(tot):.4f} median {np.median(tot):.4f} max {max(tot):.4f}")
