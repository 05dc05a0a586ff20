"""Does the kernel time of the products linearization depend on WHERE its 3 GB output lands?  Same process, same board:
fresh output buffers at shifted addresses (a dummy allocation of varying size in front), 40 launches each.
usage: python tools/placement_probe.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils
dev = torch.device("cuda:0")
S, M = 50, 100000
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device=dev)
d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
t = np.arange(S)[:, None]
us = d._us_device(np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S))
keep = []
for trial, shift_mb in enumerate((0, 1, 3, 17, 64, 257, 1000, 0, 5000, 0)):
    torch.cuda.empty_cache()
    dummy = torch.empty(shift_mb << 20, dtype=torch.uint8, device=dev) if shift_mb else None
    r = d.linearize_device(us, factored=False)
    out = {k: r[k] for k in ("G", "W", "g_up", "Z", "part", "sums") if k in r and r[k] is not None}
    for _ in range(10):
        d.linearize_device(us, factored=False, out=r)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        d.linearize_device(us, factored=False, out=r)
    b.record(); torch.cuda.synchronize()
    print("trial %d shift %5d MB  G at 0x%x  %.4f ms per launch" % (trial, shift_mb, r["G"].data_ptr(), a.elapsed_time(b) / 40))
    keep.append(dummy)
    del r, out
    if trial % 3 == 2:
        keep.clear()
