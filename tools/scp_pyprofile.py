import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from riskaversetrajopt_amd import scp, drone_risk, drone_utils, cvar_cuts, _lib
M, S = 100000, 50
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device="cuda:0")
model = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
out = scp.run_drone_reduced(model, num_scp_iters_max=40)
us = out["us"]
# line-level timing of a converged iteration with sys.setprofile-free manual stamps: wrap the pieces
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for k in range(200):
    us2, t, info = model.solve_reduced(us, 40 + k)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
