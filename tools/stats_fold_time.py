"""GPU time per call of the reduction stage of a single-GPU step: rato_sums_and_risk_stats (one call) against
rato_sum_partials + rato_risk_stats (two calls), 20 calls per captured hipGraph.  usage: python tools/stats_fold_time.py"""
import sys, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import stats
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
N = 20
for M, nblk, ncols in ((10000, 157, 126), (50000, 1176, 90), (100000, 1563, 306), (100000, 400, 306)):
    Z = 0.9 + 0.05 * torch.randn(M, generator=g, device=dev)
    part = torch.randn((nblk, ncols), generator=g, device=dev)
    ws = stats.new_workspace(M, dev)
    out = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)
    sums = torch.empty(ncols, dtype=torch.float64, device=dev)

    def fused():
        stats.sums_and_risk_stats_device(part, Z, 0.1, workspace=ws, sums_out=sums, out=out)

    def split():
        stats.sum_partials(part, out=sums)
        stats.risk_stats_device(Z, 0.1, workspace=ws, out=out)

    def only_stats():
        stats.risk_stats_device(Z, 0.1, workspace=ws, out=out)

    for name, fn in (("fused", fused), ("split", split), ("stats only", only_stats)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(N):
                fn()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            graph.replay()
        b.record(); torch.cuda.synchronize()
        print("M=%7d part %4d x %3d  %-10s %.1f us/call" % (M, nblk, ncols, name, a.elapsed_time(b) * 1000 / (10 * N)))
