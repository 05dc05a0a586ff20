"""Cuts, oracle / master / define seconds of every SCP iteration of the drone at the bench's size (one line per iteration)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from riskaversetrajopt_amd import scp, drone_risk, drone_utils
M, S = 100000, 50
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device="cuda:0")
model = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
model.solve_reduced(model.initial_guess_us_mat(), 2)
out = scp.run_drone_reduced(model, num_scp_iters_max=60)
for i in range(60):
    print(f"{i:3d} cuts {out['cuts'][i]:3d} oracle {1e3*out['oracle_s'][i]:.3f} ms solve {1e3*out['solve_s'][i]:.3f} define {1e3*out['define_s'][i]:.3f} L2 {out['L2_error'][i]:.2e}")
print("total", out['cumulative_s'][-1])
