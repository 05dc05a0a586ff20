#!/usr/bin/env python3
"""Where does drone_tail_rows_rollout_kernel (one cut) spend its time?  Needs a -DRATO_TDIAG build of the library
(RATO_SAA_LIB=/path/to/lib.so): wave 0 of every block leaves the shader-clock ticks of its phases in `part`.
    python tools/tail_phases.py [M] [S]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riskaversetrajopt_amd import _lib, drone_risk, drone_utils   # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7)
d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
dW_, mass_, Q_, _ = d._inputs(None)
cs = d._reduced_cut_solver(M, mass_.numel())
cs.implicit = None
cs.rollout = ("drone", d._params(M, mass_.numel()), dW_, mass_, Q_)
t = np.arange(S)[:, None]
u = (np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)).reshape(-1)
cs.set_linearization_point(u)
cs.evaluate(None, None, 0, None, u * 1.01, slot=0)
st = _lib.current_stream()
slots = torch.zeros(1, dtype=torch.int32, device=d.device)
part = torch.zeros((cs.nblk, cs.nc), dtype=torch.float64, device=d.device)
for _ in range(3):
    cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
torch.cuda.synchronize()
p = part.cpu().numpy()[:, :7]
names = ["statistics record, m / arg loads, compaction", "forward (rollout to t*)", "barrier", "row gradient + adjoint sweep", "barrier + write-out"]
tot = p[:, :5].sum(axis=1)
print(f"M {M} S {S}: {cs.nblk} blocks, tail samples per block mean {p[:, 5].mean():.1f} max {p[:, 5].max():.0f}")
for i, n in enumerate(names):
    print(f"  {n:48s} mean {p[:, i].mean():8.0f} ticks  ({100 * p[:, i].mean() / tot.mean():4.1f} %)")
print(f"  block total mean {tot.mean():.0f} ticks, max {tot.max():.0f};  blocks end over {(p[:, 6].max() - p[:, 6].min()) * 1e-2:.1f} us (100 MHz clock)")


def sclk_mhz():
    q = part.cpu().numpy()
    return float(np.median(q[:, :5].sum(axis=1) / np.maximum(q[:, 7], 1.0)) * 100.0)


if os.environ.get("RATO_SAA_LIB"):     # (diagnostic build) the shader clock the kernel sees, back to back and at the SCP's cadence
    import time
    for _ in range(300):
        cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
    torch.cuda.synchronize()
    print(f"shader clock seen by the kernel: {sclk_mhz():.0f} MHz after 300 launches back to back")
    for pause_us in (10, 30, 100, 1000):
        for _ in range(400):
            cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e6 < pause_us:
                pass
        print(f"                                 {sclk_mhz():.0f} MHz with a synchronisation and {pause_us} us of host time between launches")
    sys.exit(0)


def per_launch_us(n=200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
    a.record()
    for _ in range(n):
        cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


full = per_launch_us()
keep = cs.ring_res[0].clone()
cs.ring_res[0, 10] = 1e30          # t* above every m value: no sample carries a weight, the blocks do their set-up and leave
cs.ring_res[0, 8] = 0.0
cs.ring_res[0, 9] = 0.0
empty = per_launch_us()
cs.ring_res[0] = keep
print(f"back-to-back launches, HIP events: {full:.1f} us per launch with the tail; {empty:.1f} us with NO sample in the tail "
      f"(launch, statistics record, m / arg loads, compaction, write-out)")


# does a kernel cost more between OTHER kernels than between copies of itself?  (the oracle round trip alternates four)
def timed(fn, n=200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        fn()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


from riskaversetrajopt_amd import stats as rstats   # noqa: E402
cs._x_np[:] = (u * 0.01).reshape(S, 3)
_lib.copy_async(cs.x_dev, cs.x_host, st)
m_buf, arg_buf = cs.ring_m[1], cs.ring_arg[1]
ws = rstats.new_workspace(M, d.device)
out = torch.empty(rstats.N_STATS, dtype=torch.float64, device=d.device)
f_row = lambda: cs._rollout_rowmax(m_buf, arg_buf, st)
f_sel = lambda: rstats.risk_stats_device(m_buf, 0.1, workspace=ws, out=out)
f_tail = lambda: cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
t_row, t_sel, t_tail = timed(f_row), timed(f_sel), timed(f_tail)
t_all = timed(lambda: (f_row(), f_sel(), f_tail()))
print(f"each kernel 200 x back to back: rowmax {t_row:.1f}, selection {t_sel:.1f}, tail rows {t_tail:.1f} us -> sum {t_row + t_sel + t_tail:.1f}; "
      f"the three alternating: {t_all:.1f} us per round")
