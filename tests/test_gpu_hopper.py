"""GPU parity: hopper slip kernels vs the fp64 oracle and golden fixtures;
full-size properties at BASELINE config C4 (M=5e4, S=60, 40 contacts)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

H_ATOL = 2e-5      # h = fx - mu fz with fz ~ 32, mu ~ 0.1: fp32 rounding of mu*fz ~ 4e-7, 30-term sums
MU_ATOL = 2e-6


def _models(S, M, seed=1, method='saa', alpha=0.2):
    from oracle import hopper as oh
    from riskaversetrajopt_amd import hopper
    fields = oh.sample_friction_fields(np.random.RandomState(seed), M)
    return oh.Model(*fields, method=method, alpha=alpha, S=S), hopper.Model(M, method, alpha, S=S, fields=fields)


def synthetic_Z(model, seed=5):
    rng = np.random.RandomState(seed)
    S, M = model.S, model.M
    Z = np.zeros(model.num_vars)
    xs = np.zeros((S + 1, 8))
    xs[:, 0] = np.linspace(0, 0.15, S + 1)
    xs[:, 1] = 1.0
    xs[:, 2] = 0.2 * np.sin(np.linspace(0, 3, S + 1))
    xs[:, 3] = 0.9 + 0.1 * np.cos(np.linspace(0, 2, S + 1))
    us = np.zeros((S, 4))
    us[:, 3] = 32.0 + rng.randn(S)
    us[:, 2] = 0.08 * us[:, 3] + 0.3 * rng.randn(S)
    Z[:(S + 1) * 8] = xs.reshape(-1)
    Z[(S + 1) * 8:(S + 1) * 8 + S * 4] = us.reshape(-1)
    Z[(S + 1) * 8 + S * 4:-2] = 0.1 * rng.rand(M)
    Z[-2], Z[-1] = 0.03, -0.4
    return Z


def test_sampler_draw_order_matches_oracle():
    from oracle import hopper as oh
    from riskaversetrajopt_amd import hopper
    a = oh.sample_friction_fields(np.random.RandomState(1), 30)
    np.random.seed(1)
    b = hopper.sample_friction_fields(30)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("S,M", [(30, 30), (60, 700), (30, 1), (60, 257)])
@pytest.mark.parametrize("method", ["saa", "baseline"])
def test_slip_rows_vs_oracle(S, M, method):
    o, d = _models(S, M, method=method)
    Z = synthetic_Z(o)
    np.testing.assert_allclose(d.slip_risk_constraints(Z), o.slip_risk_constraints(Z), rtol=0, atol=H_ATOL)
    px, forces = o.contact_inputs(Z)
    pxd, forcesd = d.contact_inputs(Z)
    assert np.array_equal(px, pxd) and np.array_equal(forces, forcesd)
    ok_o, Z_o = o.no_slip_constraints_verification(px, forces)
    ok, Zs = d.no_slip_constraints_verification(px, forces)
    np.testing.assert_allclose(Zs, Z_o, rtol=0, atol=H_ATOL)
    assert np.all(np.abs(Z_o[ok != ok_o] - 1e-6) < 1e-4)


@pytest.mark.parametrize("S,M", [(30, 30), (60, 300)])
def test_partials_and_hessian_sums_vs_oracle(S, M):
    o, d = _models(S, M)
    Z = synthetic_Z(o)
    px, forces = o.contact_inputs(Z)
    h_o, dfz_o, dpx_o = o.slip_partials(px, forces)
    h, dfz, dpx = d.slip_partials(px, forces)
    np.testing.assert_allclose(h, h_o, rtol=0, atol=H_ATOL)
    np.testing.assert_allclose(dfz, dfz_o, rtol=0, atol=MU_ATOL)
    np.testing.assert_allclose(dpx, dpx_o, rtol=1e-5, atol=2e-5)
    lam = np.random.RandomState(2).rand(*h_o.shape)
    D1_o, D2_o = o.slip_hessian_sums(px, forces, lam)
    D1, D2 = d.slip_hessian_sums(px, forces, lam)
    np.testing.assert_allclose(D1, D1_o, rtol=2e-5, atol=1e-5 * np.sqrt(M))
    np.testing.assert_allclose(D2, D2_o, rtol=2e-5, atol=1e-4 * np.sqrt(M))


@pytest.mark.parametrize("name", ["hopper_S30_M30", "hopper_S60_M24"])
def test_golden_fixture(name):
    from riskaversetrajopt_amd import hopper
    f = np.load(os.path.join(G, name + ".npz"))
    S, M = int(f["S"]), int(f["M"])
    fields = (f["intensities"], f["thetas"], f["taus"])
    d = hopper.Model(M, 'saa', float(f["alpha"]), S=S, fields=fields)
    np.testing.assert_allclose(d.slip_risk_constraints(f["Z"]), f["gs"], rtol=0, atol=H_ATOL)
    db = hopper.Model(M, 'baseline', float(f["alpha"]), S=S, fields=fields)
    np.testing.assert_allclose(db.slip_risk_constraints(f["Z"]), f["gs_baseline"], rtol=0, atol=H_ATOL)
    h, dfz, dpx = d.slip_partials(f["px"], f["forces"])
    np.testing.assert_allclose(dfz, f["dh_dfz"], rtol=0, atol=MU_ATOL)
    np.testing.assert_allclose(dpx, f["dh_dpx"], rtol=1e-5, atol=2e-5)
    D1, D2 = d.slip_hessian_sums(f["px"], f["forces"], f["lam"])
    np.testing.assert_allclose(D1, f["D1"], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(D2, f["D2"], rtol=2e-5, atol=1e-4)
    st = d.monte_carlo_statistics(f["px"], f["forces"], alpha=0.2)
    assert abs(st["var"] - f["var"]) < 5e-5 and abs(st["cvar"] - f["avar"]) < 5e-5


# lambda-weighted Hessian sums over 5e4 samples, |D| up to 6e2 (hardware sin / cos: 1.5e-6 per term, fp32 block partials):
# measured max |err| 7.9e-5 (D1), 1.09e-2 (D2) -- profiles/r05_tolerances.txt.  D1: 3x measured (round 4: 2.2e-3);
# D2: the round-4 limit, 2.1x measured, kept (3x measured would be looser)
D1_ATOL_C4, D2_ATOL_C4 = 2.5e-4, 1e-4 * np.sqrt(50000)


def test_full_size_C4_properties():
    """M=5e4, S=60 (40 contacts; the contact-phase mask is the 'hybrid branch')."""
    from oracle import stats as ostats
    S, M = 60, 50000
    o, d = _models(S, M)
    Z = synthetic_Z(o)
    px, forces = o.contact_inputs(Z)
    assert px.shape == (40,)
    r = d.slip_device(px, forces, want_Z=True, want_h=True, want_deriv=True)
    # Z is the max over contacts of h (atomic-max path == direct path)
    assert bool((r["h"].max(dim=0).values == r["Z"]).all())
    # dh/dfz == -mu and h == fx + dh_dfz*fz  (linearity in the forces)
    import torch
    fx = torch.as_tensor(forces[:, 0], dtype=torch.float32, device=r["h"].device)[:, None]
    fz = torch.as_tensor(forces[:, 1], dtype=torch.float32, device=r["h"].device)[:, None]
    assert (r["h"] - (fx + r["dh_dfz"] * fz)).abs().max().item() < 1e-5
    # statistics against the oracle on all samples
    ok_o, Z_o = o.no_slip_constraints_verification(px, forces)
    np.testing.assert_allclose(r["Z"].cpu().numpy(), Z_o, rtol=0, atol=H_ATOL)
    st = d.monte_carlo_statistics(px, forces, alpha=0.1)
    assert abs(st["var"] - ostats.monte_carlo_var(Z_o, 0.1)) < 5e-5
    assert abs(st["cvar"] - ostats.monte_carlo_avar(Z_o, 0.1)) < 5e-5
    # the Jacobian slices IPOPT consumes (hopper.py:569, :577-580) at full size: h, dh/dfz, dh/dpx for every 997th sample
    # and all 40 contacts against the oracle, and the lambda-weighted Hessian sums over ALL samples (hopper.py:300-367)
    h_o, dfz_o, dpx_o = o.slip_partials(px, forces)                      # (M, C) each
    idx = np.arange(0, M, 997)
    tidx = torch.as_tensor(idx, device=r["h"].device)
    np.testing.assert_allclose(r["h"][:, tidx].t().cpu().numpy(), h_o[idx], rtol=0, atol=H_ATOL)
    np.testing.assert_allclose(r["dh_dfz"][:, tidx].t().cpu().numpy(), dfz_o[idx], rtol=0, atol=MU_ATOL)
    np.testing.assert_allclose(r["dh_dpx"][:, tidx].t().cpu().numpy(), dpx_o[idx], rtol=1e-5, atol=2e-5)
    lam = np.random.RandomState(2).rand(*h_o.shape)
    D1_o, D2_o = o.slip_hessian_sums(px, forces, lam)
    D1, D2 = d.slip_hessian_sums(px, forces, lam)
    from tests import _tol as tol
    tol.report("C4 D1 max |err| / (atol + rtol |ref|)", float(np.max(np.abs(D1 - D1_o) / (D1_ATOL_C4 + 2e-5 * np.abs(D1_o)))), 1.0)
    tol.report("C4 D2 max |err| / (atol + rtol |ref|)", float(np.max(np.abs(D2 - D2_o) / (D2_ATOL_C4 + 2e-5 * np.abs(D2_o)))), 1.0)
    tol.report("C4 D1 max |err|", float(np.max(np.abs(D1 - D1_o))), D1_ATOL_C4)
    tol.report("C4 D2 max |err|", float(np.max(np.abs(D2 - D2_o))), D2_ATOL_C4)
    tol.report("C4 D1 / D2 max |ref|", float(max(np.abs(D1_o).max(), np.abs(D2_o).max())), 0.0)
    np.testing.assert_allclose(D1, D1_o, rtol=2e-5, atol=D1_ATOL_C4)
    np.testing.assert_allclose(D2, D2_o, rtol=2e-5, atol=D2_ATOL_C4)
    # the reference's matrices at full size (Model.slip_jacobian / slip_hessian; 16 M structural entries written on the
    # device in CSC order): pattern identical to the oracle's assembler, values to the tolerances of the partials above
    J, J_o = d.slip_jacobian(Z), o.slip_jacobian(Z)
    assert J.shape == J_o.shape == (1 + M + M * 40 + 1, o.num_vars)
    np.testing.assert_array_equal(J.indptr, J_o.indptr)
    np.testing.assert_array_equal(J.indices, J_o.indices)
    np.testing.assert_allclose(J.data, J_o.data, rtol=1e-4, atol=3e-5)
    vals, indices, indptr, _ = d.slip_jacobian_device(Z)
    C = 40
    x = vals[:3 * C * M].view(C, 3, M)                     # [c][x0, x2, x3][i] = dh_dpx * chain: an fp32 product, exact
    Jee = torch.as_tensor(d.contact_chain(Z)[0], dtype=torch.float32, device=vals.device)
    assert torch.equal(x, r["dh_dpx"][:, None, :] * Jee[:, :, None])
    u = vals[3 * C * M:5 * C * M].view(C, 2, M)
    assert bool((u[:, 0] == 1.0).all()) and torch.equal(u[:, 1], r["dh_dfz"])
    assert bool((vals[5 * C * M:].abs() == 1.0).sum() == vals.numel() - 5 * C * M - 1)          # constants: +-1 and M alpha
    H, H_o = d.slip_hessian(Z, lam), o.slip_hessian(Z, lam)
    np.testing.assert_array_equal(H.indptr, H_o.indptr)
    np.testing.assert_array_equal(H.indices, H_o.indices)
    np.testing.assert_allclose(H.data, H_o.data, rtol=1e-4, atol=2e-5 * np.abs(H_o.data).max())


@pytest.mark.parametrize("M", [300, 50000])
def test_inputs_by_value_equal_the_staged_upload(M):
    """px / forces in the kernel's argument block (rato_hopper_slip_host_inputs, the default) and through the pinned
    upload into a device buffer (rato_hopper_slip): the same kernel arithmetic, bit for bit; the folded second stage
    of the Hessian sums (reduce=False + sums_and_risk_stats) equals the separate sum_partials."""
    import torch
    from riskaversetrajopt_amd import hopper, stats
    _, d = _models(60, M)
    C = d.time_jump + (d.S - d.time_land)
    rng = np.random.RandomState(3)
    px = np.linspace(0.0, 0.2, C)
    fz = 32.0 + rng.randn(C)
    forces = np.stack([0.08 * fz + 0.3 * rng.randn(C), fz], axis=1)
    lam = torch.rand((C, M), device="cuda")
    a = d.slip_device(px, forces, lam=lam, want_deriv=True, staged=False)
    b = d.slip_device(px, forces, lam=lam, want_deriv=True, staged=True)
    for k in ("Z", "h", "dh_dfz", "dh_dpx", "hess"):
        assert torch.equal(a[k], b[k]), k
    c = d.slip_device(px, forces, lam=lam, want_deriv=True, reduce=False)
    assert c["hess"] is None
    sums, st = stats.sums_and_risk_stats_device(c["part"], c["Z"], 0.1)
    assert torch.equal(sums, a["hess"])
    assert torch.equal(st, stats.risk_stats_device(a["Z"], 0.1))


def test_host_inputs_entry_rejects_too_many_contacts():
    import ctypes as C
    import torch
    from riskaversetrajopt_amd import _lib, hopper
    lib = _lib.load()
    n = hopper.MAX_HOST_CONTACTS + 1
    host = np.zeros(n, np.float32)
    f = torch.zeros((30, 64), device="cuda")
    Z = torch.empty(64, device="cuda")
    hp = C.c_void_p(host.ctypes.data)
    rc = lib.rato_hopper_slip_host_inputs(64, n, hp, hp, hp, _lib.ptr(f), _lib.ptr(f), _lib.ptr(f), None, _lib.ptr(Z),
                                          None, None, None, None, _lib.current_stream())
    assert rc == -1                                   # RATO_EINVAL
    rc = lib.rato_hopper_slip_host_inputs(64, n - 1, hp, hp, hp, _lib.ptr(f), _lib.ptr(f), _lib.ptr(f), None,
                                          _lib.ptr(Z), None, None, None, None, _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()


def test_reference_matrices_beyond_the_by_value_contact_limit():
    """More than RATO_HOPPER_MAX_HOST_CONTACTS (128) contact steps (S = 300: 200 contacts): the chain factors and the contact
    inputs no longer fit the kernels' argument blocks and travel through device arrays (chain_dev of
    rato_hopper_emit_jacobian_values, host_inputs = 0 of rato_hopper_slip_hessian) -- same matrices as the oracle's."""
    S, M = 300, 70
    o, d = _models(S, M)
    Z = synthetic_Z(o)
    px, _ = o.contact_inputs(Z)
    assert px.shape[0] == 200
    J, J_o = d.slip_jacobian(Z), o.slip_jacobian(Z)
    np.testing.assert_array_equal(J.indptr, J_o.indptr)
    np.testing.assert_array_equal(J.indices, J_o.indices)
    np.testing.assert_allclose(J.data, J_o.data, rtol=1e-4, atol=3e-5)
    lam = np.random.RandomState(3).rand(M, 200)
    H, H_o = d.slip_hessian(Z, lam), o.slip_hessian(Z, lam)
    np.testing.assert_array_equal(H.indptr, H_o.indptr)
    np.testing.assert_array_equal(H.indices, H_o.indices)
    np.testing.assert_allclose(H.data, H_o.data, rtol=1e-4, atol=2e-5 * np.abs(H_o.data).max())
    # 'baseline' layout through the same path
    ob, db = _models(S, M, method='baseline') if "method" in _models.__code__.co_varnames else (None, None)
    if db is not None:
        Jb, Jb_o = db.slip_jacobian(Z), ob.slip_jacobian(Z)
        np.testing.assert_array_equal(Jb.indices, Jb_o.indices)
        np.testing.assert_allclose(Jb.data, Jb_o.data, rtol=1e-4, atol=3e-5)
