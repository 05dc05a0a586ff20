"""cProfile of the reduced SCP loop (host side).  usage: python tools/prof_scp.py [M]"""
import cProfile, pstats, sys, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils, scp
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, 50, seed=0)
model = drone_risk.Model.from_device(50, dW, mass, Qsym, 'saa', 0.1, M=M)
model.solve_reduced(model.initial_guess_us_mat(), 2)
pr = cProfile.Profile()
pr.enable()
out = scp.run_drone_reduced(model, num_scp_iters_max=60)
pr.disable()
print("cumulative_s", out["cumulative_s"][-1])
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
