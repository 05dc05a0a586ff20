"""Step time (linearize + sample sums + exact VaR / CVaR) with the statistics as a launch behind the kernel and as extra
workgroups of the kernel's own launch (params.stats_*): eager and as a replayed hipGraph.
usage: python tools/fused_stats_time.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from riskaversetrajopt_amd import drone_risk, drone_utils, driving, stats   # noqa: E402


def timed(fn, n):
    for _ in range(max(3, n // 10)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for system, M, S, kw in (("drone", 100000, 50, dict(factored=False)), ("drone", 10000, 50, dict(factored=False)),
                         ("driving", 125000, 40, {}), ("driving", 10000, 40, {})):
    if system == "drone":
        dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7)
        d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
        us = np.tile([0.3, 0.05, 0.0], (S, 1))
    else:
        dW, x0, ws_, wr = driving.sample_uncertain_parameters_device(M, S, seed=7)
        d = driving.Model.from_device(S, dW, x0, ws_, wr, 'saa', 0.05)
        us = np.tile([0.1, 0.01], (S, 1))
    usd = d._us_device(us)
    ws = stats.new_workspace(M, d.device)
    st = torch.empty(stats.N_STATS, dtype=torch.float64, device=d.device)
    r, _ = d.step_device(usd, workspace=ws, stats_out=st, **kw)
    n = 200 if M <= 20000 else 60
    res = {}
    res["eager, separate launch"] = timed(lambda: d.step_device(usd, out=r, workspace=ws, stats_out=st, fused=False, **kw), n)
    res["eager, in the launch"] = timed(lambda: d.step_device(usd, out=r, workspace=ws, stats_out=st, fused=True, **kw), n)
    for name, c in (("graph, separate launch", False), ("graph, in the launch", True)):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            d.step_device(usd, out=r, workspace=ws, stats_out=st, fused=c, **kw)
        res[name] = timed(g.replay, n)
    # the kernel alone
    res["linearize alone (eager)"] = timed(lambda: d.linearize_device(usd, out=r, **(dict(reduce=False, **kw) if system == "drone" else kw)), n)
    print(f"{system} M={M} S={S}: " + "  ".join(f"{k}: {v:.1f} us" for k, v in res.items()), flush=True)
