"""GPU parity ON THE METRIC'S OWN CONFIGURATION: drone_risk, M = 1e5 samples, S = 50 (BASELINE.json `metric`), i.e.
the exact launch shape bench.py times — 1563 tiles on 512 workgroup slots, grid.y = 1, ~3 rounds of residency —
for both output representations of the row-parallel kernel (products = SURVEY 8d, factored), the generators-only
linearization and the Jacobian-free row maximum; plus one S = 20 case with n_tiles >= slots.
Reference functions: drone/drone_risk.py:239-296 (per-sample linearization + mean), :656-662 (Z), :663-695 and
drone_main_plot.py:640-652 (CVaR, VaR).  Checker: the fp64 oracle, itself pinned by tests/test_reference_pin.py."""
import numpy as np
import pytest

from tests import _tol as tol

pytestmark = pytest.mark.gpu


def graze(S):
    t = np.arange(S)[:, None]
    return np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)


_CACHE = {}


def batch(S, M):
    """(oracle model, device model, us) on identical draws; cached across the tests of this module."""
    key = (S, M)
    if key not in _CACHE:
        from oracle import drone as od
        from riskaversetrajopt_amd import drone_risk
        rng = np.random.RandomState(0)
        DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
        _CACHE.clear()
        _CACHE[key] = (od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1),
                       drone_risk.Model(S, DWs, masses, obs_Qs, 'saa', 0.1), graze(S))
    return _CACHE[key]


def oracle_means(o, us, chunk=1000):
    """mean_i of v_final_du (6,3S) and of -v_final + v_final_du.u (6,)  (drone_risk.py:271,294-296), in chunks."""
    from oracle import drone as od
    S, M = o.S, o.M
    acc, rhs = np.zeros((6, 3 * S)), np.zeros(6)
    uvec = us.reshape(-1)
    for lo in range(0, M, chunk):
        sl = slice(lo, lo + chunk)
        part = od.Model(S, o.DWs[sl], o.masses[sl], o.obs_Qs[sl])
        xs = part.us_to_state_trajectories(us)
        Phi = part.sensitivities(us, xs)
        fdu = np.zeros((xs.shape[0], 6, 3 * S))
        for a in range(3):
            fdu[:, a, a::3] = Phi[:, S, a, :, 0]
            fdu[:, 3 + a, a::3] = Phi[:, S, a, :, 1]
        acc += fdu.sum(0)
        rhs += (-(xs[:, S] - od.x_final) + fdu @ uvec).sum(0)
    return acc / M, rhs / M


def check_against_oracle(o, d, us, r, idx):
    """sampled Jacobian rows + g_up, full-batch means, Z and the statistics of one linearize result"""
    import torch
    from oracle import drone as od, stats as ostats
    S, M = o.S, o.M
    sub = od.Model(S, o.DWs[idx], o.masses[idx], o.obs_Qs[idx])
    _, _, _, gdu_o, gup_o = sub.get_all_constraints_coeffs(us)
    if r["G"] is not None:
        G_dev = d.packed_jacobian(r)                                              # (n_pairs, 2, 3, M)
        gdu = d.expand_g_obs_du(G_dev[..., torch.as_tensor(idx, device=G_dev.device)])
        tol.assert_jac_close(gdu, gdu_o, what="g_obs_du (sampled)")
        assert np.array_equal(gdu == 0.0, gdu_o == 0.0)                           # causal zeros, z column
    gup = r["g_up"][:, :, torch.as_tensor(idx, device=r["g_up"].device)].permute(2, 0, 1).double().cpu().numpy()
    tol.assert_gup_close(gup, gup_o, rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL, what="g_up")
    fdu_o, rhs_o = oracle_means(o, us)
    np.testing.assert_allclose(d.expand_final_du(r["du_sum"].cpu().numpy(), 1.0 / M), fdu_o,
                               rtol=tol.MEAN_RTOL, atol=tol.MEAN_ATOL)
    np.testing.assert_allclose(r["rhs_sum"].cpu().numpy() / M, rhs_o, rtol=tol.MEAN_RTOL, atol=2e-5)
    ok_o, Z_o = o.monte_carlo_no_collisions_constraint_verification(us)
    Z = r["Z"].double().cpu().numpy()
    np.testing.assert_allclose(Z, Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL)
    return Z, Z_o


@pytest.mark.parametrize("factored", [False, True], ids=["products", "factored"])
def test_metric_config_linearize_vs_oracle(factored):
    """the bench's launch: rows kernel, M = 1e5, S = 50, grid.y = 1"""
    import torch
    from riskaversetrajopt_amd import stats
    S, M = 50, 100000
    o, d, us = batch(S, M)
    nblk, cpt, spl, tile = d.linearize_plan(M, d._mass.numel())
    assert cpt == -1 and tile == 64 and nblk == (M + 63) // 64 == 1563            # the row-parallel kernel, 1563 tiles
    r = d.linearize_device(us, factored=factored)
    assert r["factored"] == factored
    idx = np.arange(0, M, 997)                                                    # every ~997th sample, all rounds
    Z, Z_o = check_against_oracle(o, d, us, r, idx)
    # Z of the linearize kernel == Z of the eval kernel (same formulas, two kernels)
    Z_eval, _, _ = d.eval_device(us)
    assert (Z_eval - r["Z"]).abs().max().item() <= 1e-5 * max(1.0, Z_eval.abs().max().item())
    # exact selection on the device vs np.sort of the SAME fp32 values
    st = stats.risk_stats(r["Z"], 0.1)
    Zs = np.sort(r["Z"].cpu().numpy())
    k = M - int(np.floor(0.1 * M)) - 1
    assert st["var"] == float(Zs[k])
    Z64 = Zs.astype(np.float64)
    cvar = Z64[k] + np.mean(np.maximum(Z64 - Z64[k], 0.0)) / 0.1
    assert abs(st["cvar"] - cvar) <= 1e-12 * max(1.0, abs(cvar))
    assert st["frac_satisfied"] == float(np.mean(Zs <= np.float32(1e-6)))
    # and vs the fp64 oracle's statistics
    from oracle import stats as ostats
    assert abs(st["var"] - ostats.monte_carlo_var(Z_o, 0.1)) < 2e-4 * max(1.0, abs(st["var"]))
    assert abs(st["cvar"] - ostats.monte_carlo_avar(Z_o, 0.1)) < 2e-4 * max(1.0, abs(st["cvar"]))
    # run-to-run bitwise determinism at this launch shape
    from riskaversetrajopt_amd.drone_risk import untile
    r2 = d.linearize_device(us, factored=factored)
    assert torch.equal(untile(r["G"], M), untile(r2["G"], M))      # (lanes >= M of the last tile are never written)
    assert torch.equal(r["g_up"], r2["g_up"]) and torch.equal(r["sums"], r2["sums"]) and torch.equal(r["Z"], r2["Z"])


def test_metric_config_products_equal_factored_products():
    """the two representations describe the same Jacobian: W * Phi == products, entry by entry (fp32 rounding)"""
    S, M = 50, 100000
    o, d, us = batch(S, M)
    rp = d.linearize_device(us, factored=False)
    rf = d.linearize_device(us, factored=True)
    Gp, Gf = d.packed_jacobian(rp), d.packed_jacobian(rf)
    scale = Gp.abs().amax(dim=(0, 1), keepdim=True)                               # per (obstacle, sample)
    assert ((Gp - Gf).abs() <= 2e-6 * scale + 1e-12).all()
    assert (rp["g_up"] - rf["g_up"]).abs().max().item() <= 2e-4
    assert (rp["sums"] - rf["sums"]).abs().max().item() <= 1e-6 * rp["sums"].abs().max().item()


def test_metric_config_generators_and_implicit_rowmax_vs_oracle():
    """rato_drone_linearize_generators + rato_drone_rowmax_implicit at M = 1e5, S = 50 (what the SCP block runs)"""
    import ctypes as C
    import torch
    from oracle import drone as od
    from riskaversetrajopt_amd import _lib
    S, M = 50, 100000
    o, d, us = batch(S, M)
    r = d.linearize_generators_device(us)
    idx = np.arange(0, M, 997)
    check_against_oracle(o, d, us, r, idx)
    # generators == the Jacobian kernel's tables
    rf = d.linearize_device(us, factored=True, want_A22=True)
    assert (r["W"] - rf["W"]).abs().max().item() <= 1e-5 * rf["W"].abs().max().item()
    assert ((1.0 - r["A22"][:, :2].double()) - rf["A22"].double()).abs().max().item() <= 2e-6   # generators: 1 - a22
    # m_i(u) = max_r [(G_i u)_r - g_up_ir] for a second control sequence, without reading a Jacobian
    u2 = us + 0.05 * np.sin(np.arange(S))[:, None]
    ld = d._mass.numel()
    m_out = torch.empty(ld, dtype=torch.float32, device=d.device)
    arg = torch.empty(ld, dtype=torch.int32, device=d.device)
    p = d._params(M, ld)
    u2d = torch.as_tensor(u2, dtype=torch.float64, device=d.device).contiguous()
    _lib.check(d._lib.rato_drone_rowmax_implicit(C.byref(p), _lib.ptr(d._mass), _lib.ptr(r["_A22"]), 3,
                                                 _lib.ptr(r["_W"]), _lib.ptr(r["_g_up"]), -1.0, _lib.ptr(u2d),
                                                 _lib.ptr(m_out), _lib.ptr(arg), _lib.current_stream()),
               "rato_drone_rowmax_implicit")
    sub = od.Model(S, o.DWs[idx], o.masses[idx], o.obs_Qs[idx])
    _, _, _, gdu_o, gup_o = sub.get_all_constraints_coeffs(us)
    rows = gdu_o @ u2.reshape(-1) - gup_o                                         # (n, 3, S)
    m_o = rows.reshape(len(idx), -1).max(axis=1)
    m_d = m_out[torch.as_tensor(idx, device=d.device)].double().cpu().numpy()
    np.testing.assert_allclose(m_d, m_o, rtol=0, atol=2e-4 * max(1.0, np.abs(rows).max()))
    # the delta form of the same rows, g + G (u2 - us) with the kernel writing g (rows_out = 1): the differences to the
    # fp64 oracle are those of the fp32 tables times the SMALL step, not times |G u| ~ 1e2
    rg = d.linearize_generators_device(us, rows_out=1)
    x = torch.as_tensor(u2 - us, dtype=torch.float64, device=d.device).contiguous()
    _lib.check(d._lib.rato_drone_rowmax_implicit(C.byref(p), _lib.ptr(d._mass), _lib.ptr(rg["_A22"]), 3,
                                                 _lib.ptr(rg["_W"]), _lib.ptr(rg["_g_up"]), 1.0, _lib.ptr(x),
                                                 _lib.ptr(m_out), _lib.ptr(arg), _lib.current_stream()),
               "rato_drone_rowmax_implicit")
    m_delta = m_out[torch.as_tensor(idx, device=d.device)].double().cpu().numpy()
    print("rowmax vs fp64 oracle: reference form %.2e, delta form %.2e (|rows| <= %.1f)"
          % (np.abs(m_d - m_o).max(), np.abs(m_delta - m_o).max(), np.abs(rows).max()))
    np.testing.assert_allclose(m_delta, m_o, rtol=2e-5, atol=1e-4)          # the error of the fp32 rollout's g


def test_rows_kernel_full_residency_S20():
    """S = 20: 4 workgroups per CU -> 1024 slots; M = 70,000 gives 1094 tiles >= slots (grid.y = 1, > 1 round)"""
    import torch
    S, M = 20, 70000
    o, d, us = batch(S, M)
    nblk, cpt, spl, tile = d.linearize_plan(M, d._mass.numel())
    assert cpt == -1 and nblk == 1094
    for factored in (False, True):
        r = d.linearize_device(us, factored=factored)
        check_against_oracle(o, d, us, r, np.arange(0, M, 499))
