"""Phase times of drone_tail_rows_rollout_kernel from a -DRATO_TDIAG=1 build (RATO_SAA_LIB=<that .so>): per workgroup the
100 MHz ticks at: 0 start | 1 weights known | 2 compacted | 3 first noise batch arrived | 4 forward done | 5 barrier |
6 sweep done | 7 barrier | 8 column sums done | 10 end.  Prints medians over the workgroups, relative to the earliest start."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from riskaversetrajopt_amd import _lib, drone_risk, drone_utils
M, S = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 50
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7)
d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
dW_, mass_, Q_, _ = d._inputs(None)
cs = d._reduced_cut_solver(M, mass_.numel())
cs.implicit = None
cs.rollout = ("drone", d._params(M, mass_.numel()), dW_, mass_, Q_)
t = np.arange(S)[:, None]
us = (np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)).reshape(-1)
cs.set_linearization_point(us)
cs.evaluate(None, None, 0, None, us + 0.01, slot=0)
st = _lib.current_stream()
slots = torch.zeros(1, dtype=torch.int32, device=d.device)
part = torch.zeros((cs.nblk, cs.nc), dtype=torch.float64, device=d.device)
for rep in range(3):
    cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), 1, part, st)
    torch.cuda.synchronize()
p = part.cpu().numpy()
t0 = p[:, 0].min()
names = ["start", "weights", "compacted", "noise in", "forward", "barrier", "sweep", "barrier", "col sums", "(n_tail)", "end"]
print("workgroups", cs.nblk, "span of the launch %.2f us" % ((p[:, 10].max() - t0) / 100.0))
for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10):
    v = (p[:, i] - t0) / 100.0
    print("%-10s median %6.2f us   min %6.2f  max %6.2f" % (names[i], np.median(v), v.min(), v.max()))
print("n_tail per workgroup: median %d max %d" % (np.median(p[:, 9]), p[:, 9].max()))
