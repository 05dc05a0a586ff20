"""GPU parity: driving HIP path vs the fp64 oracle and golden fixtures;
full-size properties at BASELINE config C3 (M=1e4, S=40)."""
import os

import numpy as np
import pytest

from tests import _tol as tol

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _models(S, M, seed=0, method='saa'):
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving
    samples = ocar.sample_uncertain_parameters(np.random.RandomState(seed), M, method, S)
    return ocar.Model(*samples, method=method), driving.Model(M, method, 0.05, S=S, samples=samples)


def swerve(S):
    t = np.arange(S)[:, None]
    return np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01]) * (20.0 / S)


def test_sampler_draw_order_matches_oracle():
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving
    for method in ("saa", "baseline"):
        a = ocar.sample_uncertain_parameters(np.random.RandomState(0), 6, method, 20)
        np.random.seed(0)
        m = driving.Model(6, method, 0.05)
        for x, y in zip(a, (m.states_init, m.omegas_speed, m.omegas_repulsive, m.DWs)):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("S,M", [(20, 300), (40, 257), (11, 1)])
def test_rollout_and_distances_vs_oracle(S, M):
    o, d = _models(S, M)
    us = swerve(S)
    xs_o = o.us_to_state_trajectories(us)
    np.testing.assert_allclose(d.us_to_state_trajectories(us), xs_o, rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)
    np.testing.assert_allclose(d.separation_distances_of_samples(us), o.separation_distances_at_all_times(xs_o),
                               rtol=tol.G_RTOL, atol=tol.G_ATOL)
    # the reference's signature: distances along GIVEN trajectories (driving.py:232-236), batched and one sample
    np.testing.assert_allclose(d.separation_distances_at_all_times(xs_o), o.separation_distances_at_all_times(xs_o),
                               rtol=tol.G_RTOL, atol=tol.G_ATOL)
    np.testing.assert_allclose(d.separation_distances_at_all_times(xs_o[0]),
                               o.separation_distances_at_all_times(xs_o)[0], rtol=tol.G_RTOL, atol=tol.G_ATOL)
    np.testing.assert_allclose(
        d.us_to_state_trajectory(us, o.states_init[0], o.omegas_speed[0], o.omegas_repulsive[0], o.DWs[0]),
        xs_o[0], rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)
    ok_o, Z_o = o.monte_carlo_separation_constraints_verification(us)
    ok, Z = d.monte_carlo_separation_constraints_verification(us)
    np.testing.assert_allclose(Z, Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL)
    tol.assert_satisfied_close(ok, Z_o)


@pytest.mark.parametrize("S,M,spt", [(20, 300, 0), (20, 100, 4), (20, 100, 8), (20, 100, 16), (40, 130, 0),
                                      (40, 70, 16), (33, 65, 8), (2, 5, 4), (1, 3, 0), (20, 300, -1), (40, 257, -1),
                                      (33, 65, -1), (2, 5, -1), (120, 9, 0), (130, 5, 0), (70, 200, -1), (70, 300, 8)])
def test_linearization_vs_oracle(S, M, spt):
    o, d = _models(S, M)
    us = swerve(S)
    fdu_o, flo_o, _, gdu_o, gup_o = o.get_all_constraints_coeffs(us)
    r = d.linearize_device(us, cols_per_thread=spt)
    gdu = d.expand_g_obs_du(r["G"], M)
    tol.assert_jac_close(gdu, gdu_o, rel=tol.JAC_REL_ROWMAX_DRIVING, what="g_obs_du")
    assert np.all(gdu[gdu_o == 0.0] == 0.0)
    tol.assert_gup_close(r["g_up"].t().cpu().numpy(), gup_o, rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL, what="g_up")
    tol.assert_jac_close(r["final_du"].cpu().numpy(), fdu_o[0], what="final_du")
    np.testing.assert_allclose(r["final_rhs"].cpu().numpy(), flo_o[0], rtol=1e-5, atol=5e-5)
    _, Z_o = o.monte_carlo_separation_constraints_verification(us)
    np.testing.assert_allclose(r["Z"].cpu().numpy(), Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL)


def test_variants_agree_and_are_deterministic():
    from riskaversetrajopt_amd.driving import untile
    M = 200
    _, d = _models(40, M)
    us = swerve(40)
    ref = d.linearize_device(us, cols_per_thread=4)
    b = untile(ref["G"], M).cpu().numpy()
    for spt in (4, 8, 16, -1):
        r = d.linearize_device(us, cols_per_thread=spt)
        a = untile(r["G"], M).cpu().numpy()
        assert np.all(np.abs(a - b) <= 1e-5 * np.abs(b).max(axis=(0, 1), keepdims=True) + 1e-12), spt
        assert np.array_equal(a == 0.0, b == 0.0)
        np.testing.assert_allclose(r["g_up"].cpu().numpy(), ref["g_up"].cpu().numpy(), rtol=1e-5, atol=5e-5)
        again = d.linearize_device(us, cols_per_thread=spt)
        assert bool((untile(again["G"], M) == untile(r["G"], M)).all()), spt
        assert bool((again["g_up"] == r["g_up"]).all()) and bool((again["Z"] == r["Z"]).all())


def test_single_sample_api_and_baseline():
    S = 20
    o, d = _models(S, 4)
    us = swerve(S)
    fdu_o, flo_o, _, gdu_o, gup_o = o.get_all_constraints_coeffs(us)
    i = 3
    out = d.get_all_constraints_coeffs(us, o.states_init[i], o.omegas_speed[i], o.omegas_repulsive[i], o.DWs[i])
    assert out[0].shape == (4, 2 * S) and out[3].shape == (S, 2 * S) and out[4].shape == (S,)
    tol.assert_jac_close(out[3], gdu_o[i], rel=tol.JAC_REL_ROWMAX_DRIVING, what="g_obs_du")
    tol.assert_gup_close(out[4], gup_o[i], rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL, what="g_up")
    ob, db = _models(S, 5, method='baseline')
    np.testing.assert_allclose(db.us_to_state_trajectories(us), ob.us_to_state_trajectories(us),
                               rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)


@pytest.mark.parametrize("name", ["driving_S20_M16", "driving_S40_M8"])
def test_golden_fixture(name):
    from riskaversetrajopt_amd import driving
    f = np.load(os.path.join(G, name + ".npz"))
    S, M = int(f["S"]), int(f["M"])
    d = driving.Model(M, 'saa', 0.05, S=S,
                      samples=(f["states_init"], f["omegas_speed"], f["omegas_repulsive"], f["DWs"]))
    for kind in ("init", "swerve"):
        us = f[f"{kind}_us"]
        np.testing.assert_allclose(d.us_to_state_trajectories(us), f[f"{kind}_xs"],
                                   rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)
        gdu, gup = d.get_all_constraints_coeffs_batched(us)
        tol.assert_jac_close(gdu, f[f"{kind}_g_obs_du"], rel=tol.JAC_REL_ROWMAX_DRIVING, what="g_obs_du")
        tol.assert_gup_close(gup, f[f"{kind}_g_up"], rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL, what="g_up")
        fdu, flo, _ = d.sample_means(us)
        tol.assert_jac_close(fdu, f[f"{kind}_final_du"][0], what="final_du")
        np.testing.assert_allclose(flo, f[f"{kind}_final_low"][0], rtol=1e-5, atol=5e-5)
        st = d.monte_carlo_statistics(us, alpha=0.3)
        assert abs(st["var"] - f[f"{kind}_var"]) < tol.RISK_ATOL * max(1.0, abs(f[f"{kind}_var"]))
        assert abs(st["cvar"] - f[f"{kind}_avar"]) < tol.RISK_ATOL * max(1.0, abs(f[f"{kind}_avar"]))


def test_full_size_C3_properties():
    import torch
    from oracle import driving as ocar, stats as ostats
    S, M = 40, 10000
    o, d = _models(S, M)
    us = swerve(S)
    from riskaversetrajopt_amd.driving import untile
    r = d.linearize_device(us)
    Gp = untile(r["G"], M)                                          # (n_pairs,2,M)
    _, _, g = d.eval_device(us, want_g=True)
    u = torch.as_tensor(us, dtype=torch.float32, device=g.device)
    Gu = torch.zeros_like(g)
    for t in range(1, S):
        off = t * (t - 1) // 2
        Gu[t] = (Gp[off:off + t] * u[:t, :, None]).sum(dim=(0, 1))
    tol.assert_below((r["g_up"] + g - Gu).abs().max().item(), tol.LINEARITY_ABS_DRIVING, "driving linearity |g_up + g - G.u|")
    Z_eval, _, _ = d.eval_device(us)
    assert (Z_eval - r["Z"]).abs().max().item() <= 1e-5
    ok_o, Z_o = o.monte_carlo_separation_constraints_verification(us)
    st = d.monte_carlo_statistics(us, alpha=0.05)
    assert abs(st["frac_satisfied"] - ok_o.mean()) <= np.sum(np.abs(Z_o - 1e-6) < tol.NEAR_THRESHOLD) / M + 1e-12
    assert abs(st["var"] - ostats.monte_carlo_var(Z_o, 0.05)) < 2e-4
    assert abs(st["cvar"] - ostats.monte_carlo_avar(Z_o, 0.05)) < 2e-4
    idx = np.arange(0, M, 997)
    sub = ocar.Model(o.states_init[idx], o.omegas_speed[idx], o.omegas_repulsive[idx], o.DWs[idx])
    _, _, _, gdu_o, _ = sub.get_all_constraints_coeffs(us)
    gdu = d.expand_g_obs_du(Gp[..., torch.as_tensor(idx, device=g.device)])
    tol.assert_jac_close(gdu, gdu_o, rel=tol.JAC_REL_ROWMAX_DRIVING, what="g_obs_du (sampled)")


def test_C5_shard_size_properties():
    """BASELINE config 5 is driving M = 1e6 over 8 GPUs: one rank's shard is 125,000 samples.  Size-independent
    properties at that size on one GPU (inputs drawn on the device), plus a sampled comparison with the oracle."""
    import torch
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving, stats
    from riskaversetrajopt_amd.driving import untile
    S, M = 40, 125000
    dev = torch.device("cuda:0")
    dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=11, device=dev)
    d = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)
    us = swerve(S)
    r = d.linearize_device(us)
    again = d.linearize_device(us)
    # run-to-run bitwise (the lanes of the last tile beyond M are never written: two allocations need not agree there)
    n_valid = M - (M // r["tile"]) * r["tile"]
    assert torch.equal(again["G"][:-1], r["G"][:-1]) and torch.equal(again["G"][-1, ..., :n_valid], r["G"][-1, ..., :n_valid])
    assert torch.equal(again["g_up"], r["g_up"])
    # linearity: g_up + g == G.u through the packed layout
    _, _, g = d.eval_device(us, want_g=True)
    Gp = untile(r["G"], M)
    u = torch.as_tensor(us, dtype=torch.float32, device=dev)
    Gu = torch.zeros_like(g)
    for t in range(1, S):
        off = t * (t - 1) // 2
        Gu[t] = (Gp[off:off + t] * u[:t, :, None]).sum(dim=(0, 1))
    tol.assert_below((r["g_up"] + g - Gu).abs().max().item(), tol.LINEARITY_ABS_DRIVING, "driving linearity |g_up + g - G.u|")
    Z_eval, _, _ = d.eval_device(us)
    assert (Z_eval - r["Z"]).abs().max().item() <= 1e-5
    # statistics: exact selection vs a host sort of the same fp32 Z
    Zh = r["Z"].cpu().numpy().astype(np.float64)
    st = stats.risk_stats(r["Z"], 0.05)
    srt = np.sort(Zh)
    k = M - int(np.floor(0.05 * M)) - 1
    assert st["var"] == srt[k]
    assert abs(st["cvar"] - (srt[k] + np.maximum(Zh - srt[k], 0).sum() / (0.05 * M))) < 1e-9 * max(1.0, abs(st["cvar"]))
    # sampled Jacobian rows vs the fp64 oracle on the same (device-drawn) inputs
    idx = np.arange(0, M, 9973)
    ti = torch.as_tensor(idx, device=dev)
    DWs = np.zeros((len(idx), S, 8))
    DWs[:, :, 6:8] = dW[:, :, ti].permute(2, 0, 1).double().cpu().numpy()
    from riskaversetrajopt_amd import driving_params as DP
    ego0 = np.tile(np.asarray(DP.state_init, dtype=np.float64)[:4], (len(idx), 1))
    sub = ocar.Model(np.concatenate([ego0, x0[:, ti].T.double().cpu().numpy()], axis=1),
                     ws[ti].double().cpu().numpy(), wr[ti].double().cpu().numpy(), DWs)
    _, _, _, gdu_o, gup_o = sub.get_all_constraints_coeffs(us)
    tol.assert_jac_close(d.expand_g_obs_du(Gp[..., ti]), gdu_o, rel=tol.JAC_REL_ROWMAX_DRIVING, what="g_obs_du (sampled)")
    tol.assert_gup_close(r["g_up"][:, ti].T.cpu().numpy(), gup_o.reshape(len(idx), S), rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL,
                         what="C5 shard g_up (sampled)")


@pytest.mark.parametrize("gap", [4e-3, 1.0])
def test_pedestrian_near_contact(gap):
    """The reference's singular edge: force_on_pedestrian divides by |p_ego - p_ped| (driving.py:154) and the separation
    distance differentiates the same norm (driving.py:228).  Pedestrians are placed so that the ego passes within ``gap``
    of them (4e-3: closest approach < 2e-2 -- the unit normal turns by O(1) within one step and dn/dp ~ 1/r > 100; 1.0: deep
    inside the minimum separation distance, constraint strongly violated).  HIP vs the fp64 oracle on the same numbers: finite everywhere,
    states / distances to the usual tolerances away from the singular step, Jacobian rows to 1/gap times the usual one."""
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving
    S, M = 20, 64
    rng = np.random.RandomState(3)
    x0, ws, wr, DWs = ocar.sample_uncertain_parameters(rng, M, 'saa', S)
    DWs = 0.0 * DWs                                   # deterministic approach: the pass distance is what is controlled
    us = np.zeros((S, 2))
    dt = ocar.T / S
    ego = ocar.Model(x0[:1], ws[:1], wr[:1], DWs[:1]).us_to_state_trajectories(us)[0]      # ego path (sample independent)
    t_hit = 4 + rng.randint(0, 10, M)
    side = np.where(rng.rand(M) < 0.5, -1.0, 1.0)
    x0[:, 4] = ego[t_hit, 0]
    x0[:, 5] = ego[t_hit, 1] + side * gap * (1.0 + 0.2 * rng.rand(M))
    x0[:, 6:8] = 0.0                                  # standing pedestrians (they start to move under the forces)
    ws[:] = 0.0
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    samples = (r32(x0), r32(ws), r32(wr), DWs)
    o = ocar.Model(*samples)
    d = driving.Model(M, 'saa', 0.05, S=S, samples=samples)
    xs_o = o.us_to_state_trajectories(us)
    delta = np.linalg.norm(xs_o[:, :, 0:2] - xs_o[:, :, 4:6], axis=-1)
    print(f"gap {gap}: closest approach {delta.min():.3e}")
    assert delta.min() < max(2e-2, 3 * gap)
    xs = d.us_to_state_trajectories(us)
    assert np.isfinite(xs).all()
    amp = max(1.0, 1.0 / delta.min())
    np.testing.assert_allclose(xs, xs_o, rtol=tol.STATE_RTOL * amp, atol=tol.STATE_ATOL * amp)
    r = d.linearize_device(us)
    gdu = d.expand_g_obs_du(r["G"], M)
    assert np.isfinite(gdu).all() and bool(r["g_up"].isfinite().all()) and bool(r["Z"].isfinite().all())
    _, _, _, gdu_o, gup_o = o.get_all_constraints_coeffs(us)
    tol.assert_jac_close(gdu, gdu_o, rel=tol.JAC_REL_ROWMAX_DRIVING * amp, what=f"g_obs_du, pass distance {gap}")
    assert np.all(gdu[gdu_o == 0.0] == 0.0)
    ok_o, Z_o = o.monte_carlo_separation_constraints_verification(us)
    ok, Z = d.monte_carlo_separation_constraints_verification(us)
    np.testing.assert_allclose(Z, Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL * amp)
    assert (~ok_o).all() and (~ok).all()              # everybody is hit


def test_ragged_last_tile_at_the_C5_shard_size():
    """M = 125,001: one sample beyond BASELINE C5's shard (the last 64-sample tile holds ONE sample).  The samples around
    the tile boundary against the fp64 oracle, statistics exact against a host sort, run-to-run bitwise."""
    import torch
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving, stats
    from riskaversetrajopt_amd.driving import untile
    S, M = 40, 125001
    dev = torch.device("cuda:0")
    dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=5, device=dev)
    d = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)
    us = swerve(S)
    r = d.linearize_device(us)
    idx = np.array([0, 63, 64, 124927, 124990, 124999, 125000])
    ti = torch.as_tensor(idx, device=dev)
    DWs = np.zeros((len(idx), S, 8))
    DWs[:, :, 6:8] = dW[:, :, ti].permute(2, 0, 1).double().cpu().numpy()
    ego0 = np.tile(np.asarray(ocar.state_init, dtype=np.float64)[:4], (len(idx), 1))
    sub = ocar.Model(np.concatenate([ego0, x0[:, ti].T.double().cpu().numpy()], axis=1), ws[ti].double().cpu().numpy(),
                     wr[ti].double().cpu().numpy(), DWs)
    _, _, _, gdu_o, gup_o = sub.get_all_constraints_coeffs(us)
    Gp = untile(r["G"], M)
    tol.assert_jac_close(d.expand_g_obs_du(Gp[..., ti]), gdu_o, rel=tol.JAC_REL_ROWMAX_DRIVING, what="g_obs_du (ragged tile)")
    tol.assert_gup_close(r["g_up"][:, ti].T.cpu().numpy(), gup_o.reshape(len(idx), S), rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL)
    _, Z_o = sub.monte_carlo_separation_constraints_verification(us)
    np.testing.assert_allclose(r["Z"][ti].cpu().numpy(), Z_o, rtol=tol.G_RTOL, atol=tol.G_ATOL)
    Zh = r["Z"].cpu().numpy().astype(np.float64)
    assert Zh.shape == (M,) and np.isfinite(Zh).all()
    st = stats.risk_stats(r["Z"], 0.05)
    srt = np.sort(Zh)
    assert st["var"] == srt[M - int(np.floor(0.05 * M)) - 1] and st["max"] == srt[-1]
    again = d.linearize_device(us)                     # (lanes >= M of the last tile are not written: compare the samples)
    assert bool((untile(again["G"], M) == Gp).all()) and bool((again["Z"] == r["Z"]).all())
    assert bool((again["g_up"] == r["g_up"]).all())


@pytest.mark.parametrize("M,S", [(300, 20), (10000, 40), (125001, 40), (70000, 90)])
def test_tiled_noise_gives_the_same_linearization_bit_for_bit(M, S):
    """rato_car_linearize_tiled (the noise re-tiled once per batch: one contiguous block per tile of 64 samples) against
    rato_car_linearize on the [S][2][M] array: every output identical (static grid, split tiles, tile queue, long horizon)."""
    import torch
    from riskaversetrajopt_amd import driving
    dW, x0, ws_, wr = driving.sample_uncertain_parameters_device(M, S, seed=4)
    d = driving.Model.from_device(S, dW, x0, ws_, wr, 'saa', 0.05)
    t = np.arange(S)[:, None]
    us = np.hstack([0.4 * np.cos(0.3 * t) + 0.1, 0.03 * np.sin(0.5 * t) + 0.004]) * (20.0 / S)
    assert driving.Model.TILED_NOISE
    a = d.linearize_device(us)
    try:
        driving.Model.TILED_NOISE = False
        b = d.linearize_device(us)
    finally:
        driving.Model.TILED_NOISE = True
    torch.cuda.synchronize()
    assert a["cols_per_thread"] == b["cols_per_thread"] == -1
    for k in ("g_up", "Z", "final_du", "final_rhs"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(driving.untile(a["G"], M), driving.untile(b["G"], M))
    # the tiled copy follows the noise array: a new batch in the same Model is re-tiled
    dW2 = dW * 1.5
    d2 = driving.Model.from_device(S, dW2, x0, ws_, wr, 'saa', 0.05)
    c = d2.linearize_device(us)
    assert not torch.equal(c["Z"], a["Z"])


def test_per_sample_linearization_never_reuses_another_samples_noise():
    """ADVICE r4 (high), driving twin: one-sample ``inputs`` are never served from the model's re-tiled noise cache."""
    import torch
    from riskaversetrajopt_amd import driving
    S, M = 20, 5
    o, d = _models(S, M, seed=3)
    us = swerve(S)
    Z_own = d.linearize_device(us)["Z"].clone()
    args = lambda i: (o.states_init[i], o.omegas_speed[i], o.omegas_repulsive[i], o.DWs[i])
    outs = [d.get_all_constraints_coeffs(us, *args(i)) for i in range(M)]
    g_o = o.get_all_constraints_coeffs(us)[-1]
    for i in range(M):
        np.testing.assert_allclose(outs[i][4], g_o[i], rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL)
    try:
        driving.Model.TILED_NOISE = False
        for i in range(M):
            w = d.get_all_constraints_coeffs(us, *args(i))
            for a, b in zip(outs[i], w):
                assert np.array_equal(a, b), i
    finally:
        driving.Model.TILED_NOISE = True
    assert not np.array_equal(outs[0][4], outs[1][4])
    assert torch.equal(d.linearize_device(us)["Z"], Z_own)
