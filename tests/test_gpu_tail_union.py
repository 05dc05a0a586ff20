"""GPU: the kept cuts of a subproblem re-linearized in ONE pass (drone_tail_rows_rollout_union_kernel: the union of the K
tails rolled out once per block, one adjoint sweep per cut and wave) against the same cuts re-linearized one launch each
(K = 1: drone_tail_rows_rollout_kernel).  Same rollout arithmetic, same fp64 sums in another order: 1e-12 of the row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _us(S, k):
    t = np.arange(S)[:, None]
    base = np.hstack([0.6 * np.cos(0.3 * t + 0.1 * k) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    return base * (1.0 - 0.03 * k)


@pytest.mark.parametrize("M,S,K,alpha", [(10007, 20, 5, 0.1), (100000, 50, 9, 0.1), (3000, 12, 2, 0.3), (60000, 30, 21, 0.05),
                                          (700, 20, 3, 1.0), (500, 80, 3, 0.1), (300, 66, 2, 0.2), (300, 65, 2, 0.2)])
def test_union_form_equals_one_launch_per_cut(M, S, K, alpha):
    import torch
    from riskaversetrajopt_amd import _lib, drone_risk, drone_utils, stats
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=11)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', alpha, M=M)
    dW_, mass_, Q_, _ = d._inputs(None)
    cs = d._reduced_cut_solver(M, mass_.numel())
    cs.implicit = None
    cs.rollout = ("drone", d._params(M, mass_.numel()), dW_, mass_, Q_)        # the table-free oracle of the benchmarked SCP
    assert K <= cs.keep_max and K < cs.cap - 1
    u_lin = _us(S, 0).reshape(-1)
    cs.set_linearization_point(u_lin)
    rng = np.random.default_rng(5)
    for k in range(K):                       # K evaluated cuts in ring slots 0 .. K-1 (m values, arg-max rows, statistics)
        cs.evaluate(None, None, 0, None, u_lin + 0.05 * rng.standard_normal(u_lin.size), slot=k)
    cs.set_linearization_point(_us(S, 1).reshape(-1))          # the NEW linearization point the cuts are carried to
    st = _lib.current_stream()
    slots = torch.tensor(list(range(K))[::-1], dtype=torch.int32, device=d.device)          # (any order)
    part = torch.full((cs.nblk, K * cs.nc), np.nan, dtype=torch.float64, device=d.device)
    cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots), K, part, st)
    got = part.sum(dim=0).cpu().numpy().reshape(K, cs.nc)
    ref = np.zeros_like(got)
    for k in range(K):
        one = torch.full((cs.nblk, cs.nc), np.nan, dtype=torch.float64, device=d.device)
        cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots[k:k + 1]), 1, one, st)
        ref[k] = one.sum(dim=0).cpu().numpy()
    torch.cuda.synchronize()
    assert np.isfinite(got).all() and np.abs(ref).max() > 0
    scale = np.abs(ref).max(axis=1, keepdims=True)
    err = np.abs(got - ref) / scale
    print(f"M={M} S={S} K={K}: max |union - per cut| / row max = {err.max():.2e}")
    assert err.max() < 1e-12
