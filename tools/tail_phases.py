#!/usr/bin/env python3
"""What does a launch of drone_tail_rows_rollout_kernel (one cut) cost, with and without work, and do the kernels of an oracle
round trip cost more between each other than between copies of themselves?  (The per-phase shader-clock ticks and the
clock-at-cadence measurement quoted in profiles/EXPERIMENTS.md came from a -DRATO_TDIAG build of commit 4fe294d.)
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
