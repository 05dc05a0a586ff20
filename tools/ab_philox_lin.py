"""Products / factored linearization at the metric config: noise read from HBM vs regenerated while staging (same
process, alternating).  usage: python tools/ab_philox_lin.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils
dev = torch.device("cuda:0")
S, M, seed = 50, 100000, 7
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed, device=dev)
a = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
b = drone_risk.Model.from_device(S, None, mass, Q, 'saa', 0.1, M=M, noise_seed=seed)
t = np.arange(S)[:, None]
us = a._us_device(np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S))
for fact in (False, True):
    ra = a.linearize_device(us, factored=fact)
    rb = b.linearize_device(us, factored=fact)
    assert torch.equal(ra["Z"], rb["Z"]) and torch.equal(ra["g_up"], rb["g_up"])
    for rep in range(3):
        for name, mdl, r in (("read", a, ra), ("regenerated", b, rb)):
            for _ in range(5):
                mdl.linearize_device(us, factored=fact, out=r)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                mdl.linearize_device(us, factored=fact, out=r)
            e1.record(); torch.cuda.synchronize()
            print("%-9s noise %-11s %.4f ms per launch" % ("factored" if fact else "products", name, e0.elapsed_time(e1) / 60))
    del ra, rb

from riskaversetrajopt_amd import driving
for M in (125000, 1000000):
    S = 40
    dWc, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=seed, device=dev)
    ca = driving.Model.from_device(S, dWc, x0, ws, wr, 'saa', 0.05)
    cb = driving.Model.from_device(S, None, x0, ws, wr, 'saa', 0.05, noise_seed=seed)
    tt = np.arange(S)[:, None]
    usc = ca._us_device(np.hstack([0.4 * np.cos(0.4 * tt) - 0.2, 0.05 * np.sin(0.35 * tt) + 0.01]) * 0.5)
    ra, rb = ca.linearize_device(usc), cb.linearize_device(usc)
    assert torch.equal(ra["Z"], rb["Z"]) and torch.equal(ra["g_up"], rb["g_up"])
    for rep in range(3):
        for name, mdl, r in (("read", ca, ra), ("regenerated", cb, rb)):
            for _ in range(5):
                mdl.linearize_device(usc, out=r)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                mdl.linearize_device(usc, out=r)
            e1.record(); torch.cuda.synchronize()
            print("driving M=%d noise %-11s %.4f ms per launch" % (M, name, e0.elapsed_time(e1) / 40))
    del ra, rb, ca, cb, dWc
