import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_gpu_fused_stats import _model, _us
from riskaversetrajopt_amd import stats
def replay_time(fn, n=400):
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
for system, M, S in [("drone",10000,50),("driving",10000,40)]:
    d, n_u = _model(system, M, S)
    us = torch.as_tensor(_us(S, n_u, 0), dtype=torch.float32, device=d.device)
    ws = stats.new_workspace(M, d.device); st = torch.empty(stats.N_STATS, dtype=torch.float64, device=d.device)
    bufs = {}
    t_eval = replay_time(lambda: d.eval_device(us, out=bufs))
    t_mc = replay_time(lambda: d.mc_step_device(us, workspace=ws, stats_out=st, out=bufs))
    Z = d.eval_device(us, out=bufs)[0]
    t_st = replay_time(lambda: stats.risk_stats_device(Z, d.alpha, workspace=ws, out=st))
    t_two = replay_time(lambda: (d.eval_device(us, out=bufs), stats.risk_stats_device(Z, d.alpha, workspace=ws, out=st)))
    print(f"{system} M={M} S={S}: eval only {t_eval:.1f} us | mc step (in launch) {t_mc:.1f} | stats alone {t_st:.1f} | eval + stats launch {t_two:.1f}")
    t_in = replay_time(lambda: d.mc_step_device(us, workspace=ws, stats_out=st, out=bufs, in_launch=True))
    print(f"   in-launch form {t_in:.1f}")
    for K in (1, 8, 30, 120):
        usb = us[None].repeat(K, 1, 1).contiguous()
        ob = {}
        tb = replay_time(lambda: d.eval_batch_device(usb, out=ob), n=100)
        print(f"   batch K={K}: {tb:.1f} us per call = {tb / K:.2f} us per sequence")
