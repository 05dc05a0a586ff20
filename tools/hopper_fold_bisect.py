"""Bisect: hipGraph replay time of the hopper step with the partial sums folded into the statistics launch or not."""
import sys, numpy as np, torch
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
out = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)

model.slip_device(px, forces, staged=True)          # the staging buffer exists before any capture
def v0():
    model.slip_device(px, forces, lam=lam, want_deriv=True, reduce=False, staged=False)
def v1():
    r = model.slip_device(px, forces, lam=lam, want_deriv=True)
    stats.risk_stats_device(r["Z"], 0.1, workspace=ws, out=out)
def v2():
    r = model.slip_device(px, forces, lam=lam, want_deriv=True, reduce=False)
    stats.sums_and_risk_stats_device(r["part"], r["Z"], 0.1, workspace=ws, out=out)
def v3():
    r = model.slip_device(px, forces, lam=lam, want_deriv=True, reduce=False)
    stats.sum_partials(r["part"])
    stats.risk_stats_device(r["Z"], 0.1, workspace=ws, out=out)
def v4():
    model.slip_device(px, forces, lam=lam, want_deriv=True, reduce=False)
def v5():
    model.slip_device(px, forces, lam=lam, want_deriv=True)
for rep in range(2):
  for name, fn in (("slip only, by value", v0), ("reduce in slip + stats", v1), ("fold", v2), ("slip + sum_partials + stats", v3), ("slip only (no reduce)", v4), ("slip only (reduce)", v5)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-30s %.1f us/replay" % (name, e0.elapsed_time(e1) * 1000 / 200))
