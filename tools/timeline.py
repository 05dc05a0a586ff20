"""Block timeline of drone_linearize_rows_kernel from a -DRATO_DIAG=4 build (load it with RATO_SAA_LIB=...).
usage: python tools/timeline.py [M] [S] [products|factored]"""
import sys, numpy as np, torch, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device=dev)
d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
t = np.arange(S)[:, None]
us = d._us_device(np.hstack([0.6*np.cos(0.3*t)+0.3, 0.15*np.sin(0.5*t)+0.02, 0.05*np.cos(t)])*(20/S))
fact = not (len(sys.argv) > 3 and sys.argv[3] == "products")
r = d.linearize_device(us, factored=fact)
shift = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # experiment: G base address shifted by this many bytes
if shift:
    big = torch.empty(r["G"].numel() + shift // 4, dtype=torch.float32, device=dev)
    r = dict(r, G=big[shift // 4:].view(r["G"].shape))
    print("G base address mod 4096 =", r["G"].data_ptr() % 4096)
for _ in range(3):
    r = d.linearize_device(us, out=r, factored=fact)
torch.cuda.synchronize()
raw = r["part"].cpu().numpy()                       # (nblocks, 6S+6) float32
tl = np.ascontiguousarray(raw[:, :8]).view(np.uint64).astype(np.float64)   # (nblocks, 4) ticks of 100 MHz
tl = (tl - tl[:, 0].min()) * 1e-2                   # us
start, p0, p1, end = tl.T
print("blocks %d   kernel span %.1f us" % (len(tl), end.max()))
print("start: first wave of blocks <1us: %d; percentiles 50/90/99/max: %s" % ((start < 1).sum(), np.percentile(start, [50, 90, 99, 100]).round(1)))
print("phase0 (staging)  mean %.1f  p90 %.1f  max %.1f us" % ((p0 - start).mean(), np.percentile(p0 - start, 90), (p0 - start).max()))
print("phase1 (rollout)  mean %.1f  p90 %.1f  max %.1f us" % ((p1 - p0).mean(), np.percentile(p1 - p0, 90), (p1 - p0).max()))
print("phase2 (rows)     mean %.1f  p90 %.1f  max %.1f us" % ((end - p1).mean(), np.percentile(end - p1, 90), (end - p1).max()))
print("block lifetime    mean %.1f  p10 %.1f p90 %.1f us" % ((end - start).mean(), *np.percentile(end - start, [10, 90])))
print("first phase-2 entry at %.1f us; last block start %.1f us; ends: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    p1.min(), start.max(), *np.percentile(end, [50, 90, 99, 100])))
# concurrency profile: blocks in phase 2 over time
grid = np.linspace(0, end.max(), 61)
act = [(int(((p1 <= g) & (end > g)).sum()), int(((start <= g) & (end > g)).sum())) for g in grid]
print("t(us): in-phase-2 / resident")
print("  ".join("%.0f:%d/%d" % (g, a, b) for g, (a, b) in zip(grid, act)))
# per-XCD view (workgroups are dealt round-robin over the 8 XCDs: block b and b + 8 share one)
nb = len(tl)
for x in range(8):
    sel = np.arange(nb) % 8 == x
    print("blocks = %d mod 8: n %3d  lifetime mean %.1f us  last end %.1f us  last start %.1f us" % (
        x, sel.sum(), (end - start)[sel].mean(), end[sel].max(), start[sel].max()))
