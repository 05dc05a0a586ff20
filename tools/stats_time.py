"""GPU time of rato_risk_stats per call at several M (clustered and spread values), host issue excluded: 20 calls are
captured into one hipGraph and replayed.  usage: python tools/stats_time.py   (RATO_RS_PATH=multi: the 6-launch path)"""
import sys, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import stats
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
N = 20
for M in (1000, 4096, 10000, 20000, 30000, 50000, 100000, 200000, 500000, 800000, 1000000, 1048577, 8000000):
    for name, Z in (("clustered", 0.9 + 0.05 * torch.randn(M, generator=g, device=dev)),
                    ("spread", torch.randn(M, generator=g, device=dev) * torch.exp(8 * torch.rand(M, generator=g, device=dev)))):
        ws = stats.new_workspace(M, dev)
        out = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)
        for _ in range(3):
            stats.risk_stats_device(Z, 0.1, workspace=ws, out=out)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(N):
                stats.risk_stats_device(Z, 0.1, workspace=ws, out=out)
        graph.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            graph.replay()
        b.record(); torch.cuda.synchronize()
        ref = torch.sort(Z).values[M - int(0.1 * M) - 1].item()
        print("M=%8d %-9s %.1f us/call   VaR exact: %s" % (M, name, a.elapsed_time(b) * 1000 / (10 * N), out[0].item() == ref))
