"""C4 (hopper M = 5e4, S = 60): what a replayed step costs with the statistics of step n BESIDE the kernel of step n+1
(two branches of one graph) against the serial two-node step the bench times."""
import sys, time, argparse; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from riskaversetrajopt_amd import stats
a = argparse.Namespace(S=60, M=50000, mode="linearize", alpha=0.1)
dev = torch.device("cuda:0")
work = bench.HopperWork(a, dev, seed=7)
M = work.M
ws = [stats.new_workspace(M, dev) for _ in range(2)]
so = torch.empty((2, stats.N_STATS), dtype=torch.float64, device=dev)
def kernel():
    return work.hot_kernel(reduce=False)
r0 = kernel()
r1 = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in r0.items()}      # "the previous step's outputs"
def st(r, slot):
    return stats.sums_and_risk_stats_device(r["part"], r["Z"], 0.1, workspace=ws[slot], sums_out=r.get("sums"), out=so[slot])
def replay_time(g, n=500):
    for _ in range(100): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
kernel(); st(r0, 0); st(r1, 1); torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    r = kernel(); st(r, 0)
print("serial step (kernel -> sums + statistics): %.1f us" % replay_time(g1))
gk = torch.cuda.CUDAGraph()
with torch.cuda.graph(gk):
    kernel()
print("kernel alone: %.1f us" % replay_time(gk))
gs = torch.cuda.CUDAGraph()
with torch.cuda.graph(gs):
    st(r1, 1)
print("sums + statistics alone: %.1f us" % replay_time(gs))
side = torch.cuda.Stream()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        st(r1, 1)
    kernel()
    cur.wait_stream(side)
print("two branches (kernel of step n+1 || statistics of step n): %.1f us" % replay_time(g2))
