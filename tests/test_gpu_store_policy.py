"""GPU: the store policy of the row-parallel linearize kernels at its edges.  The launcher writes the Jacobian with
streaming (non-temporal) stores when the output is >= 256 MB and the batch's noise <= 128 MB
(rato_*_rows_streaming_stores; drone.hip / driving.hip), with ordinary stores otherwise -- two different store
instructions in the kernel.  On BOTH sides of BOTH edges: sampled Jacobian rows, g_up and Z against the fp64 oracle
(drone_risk.py:239-296, driving.py:260-313) on the batch's own fp32 inputs, and the first / last tiles of the output
identical between the two store forms' neighbours' arithmetic (run-to-run determinism)."""
import numpy as np
import pytest

from tests import _tol as tol

pytestmark = pytest.mark.gpu


def _idx(M, n=24):
    idx = np.unique(np.concatenate([np.linspace(0, M - 1, n).astype(np.int64), [0, 1, 63, 64, M - 65, M - 64, M - 1]]))
    return idx[(idx >= 0) & (idx < M)]


@pytest.mark.parametrize("M,streaming", [(8700, False), (8710, True), (213300, True), (213400, False)])
def test_drone_products_on_both_sides_of_the_policy_edges(M, streaming):
    import torch
    from oracle import drone as od
    from riskaversetrajopt_amd import drone_risk, drone_utils
    S = 50
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=21)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
    assert bool(d._lib.rato_drone_rows_streaming_stores(M, S, 0)) == streaming
    assert (M * 1225 * 24 >= 256e6 and M * S * 12 <= 128e6) == streaming            # the documented rule
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    r = d.linearize_device(us, factored=False)
    assert r["cols_per_thread"] == -1 and r["tile"] == 64
    idx = _idx(M)
    it = torch.as_tensor(idx, device=dW.device)
    # the batch's own inputs, as the kernels read them (fp32), in the reference's layouts
    DWs = np.zeros((idx.size, S, 6))
    DWs[:, :, 3:6] = dW[:, :, it].permute(2, 0, 1).double().cpu().numpy()
    masses = mass[it].double().cpu().numpy()
    q = Q[:, :, it].double().cpu().numpy()                                            # (n_obs, 3, n): Q00, Q01 + Q10, Q11
    obs_Qs = np.zeros((idx.size, 3, 3, 3))
    obs_Qs[:, :, 0, 0], obs_Qs[:, :, 1, 1] = q[:, 0].T, q[:, 2].T
    obs_Qs[:, :, 0, 1] = obs_Qs[:, :, 1, 0] = 0.5 * q[:, 1].T
    obs_Qs[:, :, 2, 2] = 1.0
    sub = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
    _, _, _, gdu_o, gup_o = sub.get_all_constraints_coeffs(us)
    G = r["G"]                                                                         # [n_tiles][n_pairs][2][n_obs][64]
    Gs = G[it // 64, :, :, :, it % 64].permute(1, 2, 3, 0)                             # (n_pairs, 2, n_obs, n)
    gdu = d.expand_g_obs_du(Gs)
    tol.assert_jac_close(gdu, gdu_o, what=f"g_obs_du (M = {M})")
    assert np.array_equal(gdu == 0.0, gdu_o == 0.0)
    gup = r["g_up"][:, :, it].permute(2, 0, 1).double().cpu().numpy()
    tol.assert_gup_close(gup, gup_o, rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL)
    _, Z_o = sub.monte_carlo_no_collisions_constraint_verification(us)
    np.testing.assert_allclose(r["Z"][it].double().cpu().numpy(), Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL)
    # every structural entry was written (a second launch into the same buffers leaves the same bits; ends of the buffer)
    r2 = d.linearize_device(us, factored=False)
    for sl in (slice(0, 2), slice(-3, -1)):
        assert torch.equal(r["G"][sl], r2["G"][sl])
    assert torch.isfinite(r["G"][-2]).all()


@pytest.mark.parametrize("M,streaming", [(41000, False), (41050, True), (399900, True), (400100, False)])
def test_driving_on_both_sides_of_the_policy_edges(M, streaming):
    import torch
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving
    from riskaversetrajopt_amd import driving_params as P
    S = 40
    dW, x0, ws_, wr = driving.sample_uncertain_parameters_device(M, S, seed=22)
    d = driving.Model.from_device(S, dW, x0, ws_, wr, 'saa', 0.05)
    assert bool(d._lib.rato_car_rows_streaming_stores(M, S)) == streaming
    assert (M * 780 * 8 >= 256e6 and M * S * 8 <= 128e6) == streaming
    t = np.arange(S)[:, None]
    us = np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01]) * (20.0 / S)
    r = d.linearize_device(us)
    assert r["cols_per_thread"] == -1
    idx = _idx(M)
    it = torch.as_tensor(idx, device=dW.device)
    DWs = np.zeros((idx.size, S, 8))
    DWs[:, :, 6:8] = dW[:, :, it].permute(2, 0, 1).double().cpu().numpy()
    states = np.tile(np.asarray(P.state_init, dtype=np.float64), (idx.size, 1))
    states[:, 4:8] = x0[:, it].t().double().cpu().numpy()
    sub = ocar.Model(states, ws_[it].double().cpu().numpy(), wr[it].double().cpu().numpy(), DWs, method='saa')
    _, _, _, gdu_o, gup_o = sub.get_all_constraints_coeffs(us)
    Gs = r["G"][it // 64, :, :, it % 64].permute(1, 2, 0)                              # (n_pairs, 2, n)
    gdu = d.expand_g_obs_du(Gs)
    tol.assert_jac_close(gdu, gdu_o, rel=tol.JAC_REL_ROWMAX_DRIVING, what=f"g_obs_du (M = {M})")
    tol.assert_gup_close(r["g_up"][:, it].t().double().cpu().numpy(), gup_o, rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL)
    _, Z_o = sub.monte_carlo_separation_constraints_verification(us)
    np.testing.assert_allclose(r["Z"][it].double().cpu().numpy(), Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL)
    r2 = d.linearize_device(us)
    for sl in (slice(0, 2), slice(-3, -1)):
        assert torch.equal(r["G"][sl], r2["G"][sl])
    assert torch.isfinite(r["G"][-2]).all()


@pytest.mark.parametrize("factored", [False, True])
def test_linearizations_on_two_alternating_streams_equal_the_one_stream_result(factored):
    """Consecutive linearize launches issued on two alternating streams (bench.py's `two_streams` block; also what
    dist.PipelinedSteps relies on) overlap on the chip: every stream has its own tile queue and the caller its own
    output slot per stream, so both slots must hold, bit for bit, what one launch on one stream writes."""
    import torch
    from riskaversetrajopt_amd import drone_risk, drone_utils
    M, S = 100000, 50
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    ref = d.linearize_device(us, factored=factored)
    torch.cuda.synchronize()
    from riskaversetrajopt_amd.drone_risk import untile
    keys = [k for k in ("G", "g_up", "Z", "W", "part") if isinstance(ref.get(k), torch.Tensor)]
    valid = lambda k, r: untile(r[k], M) if k == "G" else (r[k][..., :M] if k == "W" else r[k])   # (not the padding lanes)
    want = {k: valid(k, ref).clone() for k in keys}
    outs = [d.linearize_device(us, factored=factored), d.linearize_device(us, factored=factored)]
    torch.cuda.synchronize()
    for o in outs:
        for k in keys:
            o[k].zero_()
    two = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for i in range(12):
        with torch.cuda.stream(two[i & 1]):
            d.linearize_device(us, out=outs[i & 1], factored=factored)
    torch.cuda.synchronize()
    for o in outs:
        for k in keys:
            assert torch.equal(valid(k, o), want[k]), k
