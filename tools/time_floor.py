import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from riskaversetrajopt_amd import stats, _lib
lib = _lib.load()
def replay_time(fn, n=500):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
dev = torch.device("cuda:0")
ws = stats.new_workspace(1000, dev)
x = torch.zeros(64, device=dev)
one = lambda: _lib.check(lib.rato_risk_stats_init(_lib.ptr(ws), ws.numel(), _lib.current_stream()), "init")
print("1 tiny kernel per graph: %.1f us" % replay_time(one))
print("2 tiny kernels per graph: %.1f us" % replay_time(lambda: (one(), one())))
print("4 tiny kernels per graph: %.1f us" % replay_time(lambda: (one(), one(), one(), one())))
print("torch fill 64 floats per graph: %.1f us" % replay_time(lambda: x.fill_(1.0)))
for M in (1000, 10000):
    Z = torch.randn(M, device=dev) - 3.0
    st = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)
    print(f"rs_small M={M}: %.1f us" % replay_time(lambda: stats.risk_stats_device(Z, 0.1, workspace=ws, out=st)))
# eager back-to-back with events
Z = torch.randn(10000, device=dev) - 3.0
st = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(20_000_000)
a.record()
for _ in range(200): stats.risk_stats_device(Z, 0.1, workspace=ws, out=st)
b.record(); torch.cuda.synchronize()
print("rs_small M=10000 eager back to back behind a spin: %.2f us per launch" % (a.elapsed_time(b) * 1e3 / 200))
torch.cuda._sleep(20_000_000)
a.record()
for _ in range(200): one()
b.record(); torch.cuda.synchronize()
print("tiny kernel eager back to back behind a spin: %.2f us per launch" % (a.elapsed_time(b) * 1e3 / 200))
