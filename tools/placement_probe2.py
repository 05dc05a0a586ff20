"""Which buffer's placement moves the kernel time?  Re-allocate only G (3 GB), or only the small outputs, between
timings.  usage: python tools/placement_probe2.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils, _lib
dev = torch.device("cuda:0")
S, M = 50, 100000
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device=dev)
d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
t = np.arange(S)[:, None]
us = d._us_device(np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S))
def timeit(r, n=40):
    for _ in range(5):
        d.linearize_device(us, factored=False, out=r)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        d.linearize_device(us, factored=False, out=r)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
r = d.linearize_device(us, factored=False)
print("initial                     G 0x%x  %.4f ms" % (r["G"].data_ptr(), timeit(r)))
print("again (nothing changed)     G 0x%x  %.4f ms" % (r["G"].data_ptr(), timeit(r)))
for trial in range(6):
    shape, dt = r["G"].shape, r["G"].dtype
    r["G"] = None
    torch.cuda.empty_cache()
    pad = torch.empty((trial * 37 + 1) << 20, dtype=torch.uint8, device=dev)
    r["G"] = _lib.packed_buffer(shape, dev)
    print("new G only (pad %4d MB)     G 0x%x  %.4f ms" % (trial * 37 + 1, r["G"].data_ptr(), timeit(r)))
    del pad
for trial in range(4):
    for k in ("g_up", "Z", "part"):
        r[k] = torch.empty_like(r[k])
    print("new small outputs only      G 0x%x  %.4f ms" % (r["G"].data_ptr(), timeit(r)))
dW2 = dW.clone()
d2 = drone_risk.Model.from_device(S, dW2, mass, Q, 'saa', 0.1, M=M)
d = d2
print("new noise buffer            G 0x%x  %.4f ms" % (r["G"].data_ptr(), timeit(r)))
