"""Monte-Carlo validation steps (rollout -> max -> fraction / VaR / AVaR, M = 1e4) issued EAGERLY back to back from Python
against the replayed two-node graph: wall-clock per step over 2000 steps; plus the host's own time per eager call."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from tests.test_gpu_fused_stats import _model, _us
from riskaversetrajopt_amd import stats
for system, M, S in [("drone", 10000, 50), ("driving", 10000, 40)]:
    d, n_u = _model(system, M, S)
    us = torch.as_tensor(_us(S, n_u, 0), dtype=torch.float32, device=d.device)
    ws = stats.new_workspace(M, d.device); st = torch.empty(stats.N_STATS, dtype=torch.float64, device=d.device)
    bufs = {}
    step = lambda: d.mc_step_device(us, workspace=ws, stats_out=st, out=bufs)
    for _ in range(200): step()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / n * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(100): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / n * 1e6
    print(f"{system} M={M} S={S}: eager {t_eager:.1f} us per step (host issue {t_host:.1f}) | replayed graph {t_graph:.1f}")
    step_in = lambda: d.mc_step_device(us, workspace=ws, stats_out=st, out=bufs, in_launch=True)
    for _ in range(200): step_in()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step_in()
    torch.cuda.synchronize()
    print(f"    statistics in the rollout's own launch, eager: {(time.perf_counter() - t0) / n * 1e6:.1f} us per step")
