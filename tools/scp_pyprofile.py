import sys; sys.path.insert(0, '.')
import cProfile, pstats, numpy as np, torch
from riskaversetrajopt_amd import scp, drone_risk, drone_utils
M, S = 100000, 50
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device="cuda:0")
model = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
model.solve_reduced(model.initial_guess_us_mat(), 2)
scp.run_drone_reduced(model, num_scp_iters_max=60)
pr = cProfile.Profile(); pr.enable()
scp.run_drone_reduced(model, num_scp_iters_max=60)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
