"""Does solving the LATE subproblems of the SCP to the master's own accuracy (the stall rule alone stops the loop) remove the
2-cycles of amplitude ~3e-6 that about one batch in twelve ends in, and what does it cost?  Schedule: the default cut
tolerance until the SCP's own L2 change falls below `thr`, then `tight`.   usage: python tools/scp_adaptive.py [seeds...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from riskaversetrajopt_amd import drone_risk, drone_utils, scp   # noqa: E402

seeds = [int(a) for a in sys.argv[1:]] or [6, 7, 8, 3]
M, S = 100000, 50
for thr, tight in ((0.0, 1e-9), (1e-4, 1e-11), (1e-5, 1e-11), (1e-4, 1e-12), (1e-3, 1e-11)):
    tot = []
    for seed in seeds:
        dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
        d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
        d.solve_reduced(d.initial_guess_us_mat(), 2)
        d._cut_solver = None
        us_prev = d.initial_guess_us_mat()
        err, cuts, total = [], 0, 0.0
        for k in range(60):
            tol = tight if (err and err[-1] < thr) else 1e-9
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            us, _, info = d.solve_reduced(us_prev, k, tol=tol)
            torch.cuda.synchronize()
            total += time.perf_counter() - t0
            cuts += info["cuts"]
            err.append(scp.L2_error_us(us, us_prev))
            us_prev = us
        tot.append(total)
        print(f"tight {tight:.0e} below {thr:.0e} seed {seed:2d}: {total:.4f} s  cuts {cuts:4d}  L2 last {err[-1]:.1e}  (L2 at 30 / 45: {err[30]:.1e} / {err[45]:.1e})", flush=True)
    print(f"tight {tight:.0e} below {thr:.0e}: median {np.median(tot):.4f}")
