"""How long does the HOST take to issue one bench step (no GPU wait)?  usage: python tools/cpu_issue.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from riskaversetrajopt_amd import stats
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda:0")
work = bench.DroneWork(args, dev, 7)
ws = stats.new_workspace(work.M, dev)
out = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)
for _ in range(5):
    r = work.hot_kernel(); stats.risk_stats_device(r["Z"], 0.1, workspace=ws, out=out)
torch.cuda.synchronize()
for name, fn in (("linearize_device", lambda: work.hot_kernel()),
                 ("risk_stats_device", lambda: stats.risk_stats_device(r["Z"], 0.1, workspace=ws, out=out))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-20s host issue %.1f us/call, with GPU drain %.1f us/call" % (name, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
