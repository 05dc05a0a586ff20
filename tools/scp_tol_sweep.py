"""What the cutting-plane tolerance costs: the drone SCP at the bench's size with the loop's stopping violation at
1e-9 (default) ... 1e-5 on EVERY subproblem; cuts, seconds, and the distance of the 60th iterate from the default's."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from riskaversetrajopt_amd import scp, drone_risk, drone_utils
M, S = 100000, 50
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device="cuda:0")
model = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
model.solve_reduced(model.initial_guess_us_mat(), 2)
solve = model.solve_reduced
ref = None
for tol in [1e-9, 1e-8, 1e-7, 1e-6, 1e-5]:
    model.solve_reduced = lambda us, it, _t=tol, **kw: solve(us, it, tol=_t, **kw)
    out = scp.run_drone_reduced(model, num_scp_iters_max=60)
    us = np.asarray(out["us"])
    if ref is None:
        ref = us
    d = np.linalg.norm(us - ref) / np.linalg.norm(ref)
    print(f"tol {tol:.0e}: cuts {int(out['cuts'].sum())} cumulative {out['cumulative_s'][-1]:.4f} s  oracle {out['oracle_s'].sum():.4f}  "
          f"L2_last {out['L2_error'][-1]:.2e}  |us - us(1e-9)| / |us| {d:.2e}   cuts/iter {list(out['cuts'][:30])}")
