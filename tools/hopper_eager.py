"""Eager (IPOPT-callback style) cost of one hopper slip evaluation at C4 size: wall clock per call with the per-contact
inputs by value in the kernel arguments vs through the pinned staging upload.  usage: python tools/hopper_eager.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import hopper, stats
dev = torch.device("cuda:0")
M, S = 50000, 60
a, th, tau = hopper.sample_friction_fields_device(M, seed=1, device=dev)
model = hopper.Model.from_device(a, th, tau, 'saa', 0.1, S=S)
tj, tl = hopper.phase_times(S)
C = tj + (S - tl)
rng = np.random.RandomState(5)
px = np.linspace(0.0, 0.2, C)
fz = 32.0 + rng.randn(C)
forces = np.stack([0.08 * fz + 0.3 * rng.randn(C), fz], axis=1)
lam = torch.rand((C, M), device=dev)
ws = stats.new_workspace(M, dev)
for staged in (False, True, False, True):
    for what in ("async issue", "call + Z statistics read back"):
        n = 300
        for _ in range(20):
            model.slip_device(px, forces, lam=lam, want_deriv=True, staged=staged)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            r = model.slip_device(px + 1e-6 * i, forces, lam=lam, want_deriv=True, staged=staged, reduce=False)
            if what != "async issue":
                sums, st = stats.sums_and_risk_stats_device(r["part"], r["Z"], 0.1, workspace=ws)
                st.cpu()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("staged=%-5s %-32s %.1f us per call" % (staged, what, dt * 1e6))
