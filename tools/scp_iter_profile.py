"""Where a converged SCP iteration of the drone (M = 1e5, S = 50) spends its host time: the two native calls
(rato_cut_define_drone, rato_cut_solve with its own oracle / master split) and the interpreter around them."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from riskaversetrajopt_amd import scp, drone_risk, drone_utils
M, S = 100000, 50
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device="cuda:0")
model = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
us, _, _ = model.solve_reduced(model.initial_guess_us_mat(), 2)
T = {"define": [], "solve": []}
lib = model._lib
for name in ("rato_cut_define_drone", "rato_cut_solve"):
    f = getattr(lib, name)
    def wrap(*a, _f=f, _k=name.split("_")[2]):
        t0 = time.perf_counter(); r = _f(*a); T[_k].append(time.perf_counter() - t0); return r
    setattr(lib, name, wrap)
cs_lib = None
us = model.initial_guess_us_mat()
rows = []
for k in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n0 = (len(T["define"]), len(T["solve"]))
    us, t, info = model.solve_reduced(us, k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((k, info["cuts"], 1e6 * (t1 - t0), 1e6 * sum(T["define"][n0[0]:]), 1e6 * sum(T["solve"][n0[1]:]), 1e6 * info["oracle_s"], 1e6 * info["master_s"], 1e6 * (t2 - t1)))
for r in rows:
    print("it %2d cuts %2d | solve_reduced %7.1f us = define call %6.1f + solve call %7.1f (oracle %7.1f master %6.1f) + python %6.1f | final sync %5.1f" %
          (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[2] - r[3] - r[4], r[7]))
