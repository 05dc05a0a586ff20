"""Soak of the reduced SCP (generators-only linearization, implicit oracle, recycling) at random sizes: every run must
finish, stay finite and reach a small last L2 change.  usage: python tools/soak_scp.py [seconds]"""
import sys, time, numpy as np, torch, faulthandler
faulthandler.dump_traceback_later(900, exit=True)
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils, driving, scp
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(1)
t_end = time.time() + budget
n = 0
worst = 0.0
while time.time() < t_end:
    S = int(rng.choice([10, 20, 30, 50]))
    M = int(rng.choice([200, 1000, 5000, 8192, 8193, 20000, 100000, 300000]))
    alpha = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
    seed = int(rng.randint(1 << 30))
    if rng.rand() < 0.7:
        dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
        model = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', alpha, M=M)
        out = scp.run_drone_reduced(model, num_scp_iters_max=25)
    else:
        dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=seed)
        model = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', alpha)
        out = scp.run_driving_reduced(model, num_scp_iters_max=12)
    assert np.isfinite(out["us"]).all() and np.isfinite(out["t_risk"]), (S, M, alpha, seed)
    worst = max(worst, float(out["L2_error"][-1]))
    n += 1
    del model
print("soak ok: %d SCP runs, worst last L2 change %.2e" % (n, worst))
