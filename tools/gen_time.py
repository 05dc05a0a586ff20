"""Kernel time of the generators-only linearization vs the factored Jacobian kernel.  usage: python tools/gen_time.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils
dev = torch.device("cuda:0")
for M in (100000, 1000000, 10000000):
    S = 50
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=1, device=dev)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
    t = np.arange(S)[:, None]
    us = d._us_device(np.hstack([0.6*np.cos(0.3*t)+0.3, 0.15*np.sin(0.5*t)+0.02, 0.05*np.cos(t)])*(20/S))
    r = d.linearize_generators_device(us)
    for _ in range(3):
        r = d.linearize_generators_device(us, out=r)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        r = d.linearize_generators_device(us, out=r)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    bytes_ = M * 4 * (3 * S + 1 + 9 + 1) + M * 4 * (3 * S + 6 * S + 3 * S)     # inputs | A22, W, g_up
    print("M=%8d  %.3f ms per call (kernel + sum_partials)  %.0f GB/s of %.1f B per sample-step" % (M, ms, bytes_ / ms / 1e6, bytes_ / M / S))
    del d, dW, r
    torch.cuda.empty_cache()
