"""GPU: statistics in the producer's own launch (rato_saa.h: params.stats_*).  A row-parallel linearize launch carries a
few extra workgroups that wait until every tile's Z has been counted in and then run the exact selection on it -- while
the Jacobian is still being stored.  Against the same kernel followed by rato_risk_stats: every output of the kernel and
every exactly defined entry of the record (VaR, counts, max, rank, threshold) bit for bit, the fp64 sums to summation
order; eager and replayed from a hipGraph; the signal words back at zero after every launch."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXACT = [0, 2, 4, 5, 7, 8, 9, 10]          # var, frac_satisfied, max, count_satisfied, rank, count_above, count_at, t_star
SUMS = [1, 3, 6]                           # cvar, mean, tail_sum (fp64 sums: equal to summation order)


def _signal_words(ws):
    import torch
    # the signal words are the last 8 words of the workspace struct (csrc/rato_select.h: Workspace::sig)
    return ws[-32:].view(torch.int32).cpu().numpy()


def _us(S, n_u, k):
    t = np.arange(S)[:, None]
    base = np.hstack([0.6 * np.cos(0.3 * t + 0.1 * k) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    return base[:, :n_u] * (1.0 - 0.03 * k)


def _model(system, M, S):
    from riskaversetrajopt_amd import drone_risk, drone_utils, driving
    if system == "drone":
        dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=3)
        return drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M), 3
    dW, x0, ws_, wr = driving.sample_uncertain_parameters_device(M, S, seed=3)
    return driving.Model.from_device(S, dW, x0, ws_, wr, 'saa', 0.05), 2


@pytest.mark.parametrize("system,M,S", [("drone", 300, 20), ("drone", 3000, 20), ("drone", 10000, 50), ("drone", 12289, 20),
                                        ("drone", 40000, 20), ("drone", 100003, 20), ("drone", 100000, 50),
                                        ("driving", 3000, 20), ("driving", 10000, 40), ("driving", 125001, 40),
                                        ("driving", 500000, 10)])
def test_statistics_in_the_launch_equal_the_separate_launch(system, M, S):
    import torch
    from riskaversetrajopt_amd import stats
    d, n_u = _model(system, M, S)
    ws_a, ws_b = stats.new_workspace(M, d.device), stats.new_workspace(M, d.device)
    ra = rb = None
    for k in range(5):
        us = _us(S, n_u, k)
        ra, sa = d.step_device(us, out=ra, workspace=ws_a, fused=False)
        rb, sb = d.step_device(us, out=rb, workspace=ws_b, fused=True)
        torch.cuda.synchronize()
        a, b = sa.cpu().numpy(), sb.cpu().numpy()
        assert np.array_equal(a[EXACT], b[EXACT]), (k, a, b)
        np.testing.assert_allclose(b[SUMS], a[SUMS], rtol=1e-12, atol=1e-300)
        assert torch.equal(ra["Z"], rb["Z"]) and torch.equal(ra["g_up"], rb["g_up"])
        if system == "drone":
            assert torch.equal(ra["sums"], rb["sums"])
            assert torch.equal(d.packed_jacobian(ra), d.packed_jacobian(rb)) if M <= 12289 else True
        assert not _signal_words(ws_b).any()                      # counter and flag lowered again
    Zh = np.sort(rb["Z"].double().cpu().numpy())
    assert b[0] == Zh[M - int(np.floor(d.alpha * M)) - 1] and b[4] == Zh[-1]


@pytest.mark.parametrize("system,M,S", [("drone", 10000, 50), ("driving", 10000, 40), ("drone", 60000, 20)])
def test_statistics_in_the_launch_replayed_from_a_hipgraph(system, M, S):
    import torch
    from riskaversetrajopt_amd import stats
    d, n_u = _model(system, M, S)
    graphs = {}
    for fused in (False, True):
        us = torch.zeros((S, n_u), dtype=torch.float32, device=d.device)
        ws = stats.new_workspace(M, d.device)
        st = torch.empty(stats.N_STATS, dtype=torch.float64, device=d.device)
        r, _ = d.step_device(us, workspace=ws, stats_out=st, fused=fused)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            d.step_device(us, out=r, workspace=ws, stats_out=st, fused=fused)
        graphs[fused] = (g, us, st, ws, r)          # (r: the captured launches write into these buffers)
    for k in range(6):
        u = torch.as_tensor(_us(S, n_u, k), dtype=torch.float32, device=d.device)
        for fused in (False, True):
            g, us, st, ws, _ = graphs[fused]
            us.copy_(u)
            g.replay()
        torch.cuda.synchronize()
        a, b = graphs[False][2].cpu().numpy(), graphs[True][2].cpu().numpy()
        assert np.array_equal(a[EXACT], b[EXACT]), (k, a, b)
        np.testing.assert_allclose(b[SUMS], a[SUMS], rtol=1e-12, atol=1e-300)
        assert not _signal_words(graphs[True][3]).any()
    for fused in (False, True):
        g = graphs[fused][0]
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        print(f"{system} M={M} S={S} {'statistics in the launch' if fused else 'kernel + statistics launch'}: "
              f"{(time.perf_counter() - t0) / 200 * 1e6:.1f} us per replayed step")


def test_refused_where_it_cannot_work():
    import torch
    from riskaversetrajopt_amd import _lib, stats
    d, _ = _model("drone", 2000, 20)
    ws = stats.new_workspace(2000, d.device)
    out = torch.empty(stats.N_STATS, dtype=torch.float64, device=d.device)
    with pytest.raises(_lib.RatoError):                              # the column kernel carries no statistics workgroups
        d.linearize_device(_us(20, 3, 0), cols_per_thread=8, samples_per_lane=1, factored=False,
                           stats_request=(ws, out, 0.1))
    with pytest.raises(_lib.RatoError):                              # Z must be requested
        d.linearize_device(_us(20, 3, 0), want_Z=False, stats_request=(ws, out, 0.1))
    # a batch beyond one round of workgroup slots: the same call issues rato_risk_stats behind the kernel instead
    big, _ = _model("driving", 200000, 10)
    assert not big._lib.rato_car_stats_in_launch(200000, 10) and d._lib.rato_drone_stats_in_launch(2000, 20)
    ws2, out2 = stats.new_workspace(200000, big.device), torch.empty(stats.N_STATS, dtype=torch.float64, device=big.device)
    r = big.linearize_device(_us(10, 2, 0), stats_request=(ws2, out2, 0.05))
    ref = stats.risk_stats_device(r["Z"], 0.05)
    torch.cuda.synchronize()
    assert torch.equal(out2, ref)


# ---- round 5: the Monte-Carlo step (rollout -> Z -> statistics) as one call / one launch --------------------------------
@pytest.mark.parametrize("system,M,S", [("drone", 1, 20), ("drone", 300, 20), ("drone", 10000, 50), ("drone", 12289, 33),
                                        ("drone", 50000, 50), ("drone", 70001, 20), ("drone", 4097, 126),
                                        ("drone", 100, 1), ("drone", 130, 64), ("drone", 130, 65), ("drone", 70, 129),
                                        ("driving", 1, 20), ("driving", 257, 20), ("driving", 10000, 40),
                                        ("driving", 50000, 40), ("driving", 70001, 17), ("driving", 3000, 90),
                                        ("driving", 100, 1), ("driving", 130, 64), ("driving", 130, 65), ("driving", 70, 129)])
def test_tiled_eval_equals_the_plain_kernel_and_its_statistics_the_separate_launch(system, M, S):
    """rato_*_eval without trajectories runs the tiled kernel (one wave per 64 samples, noise batches in flight, the ego
    tables folded in the launch for driving); with trajectories the plain one.  Z and g agree to the bit; the record the
    launch leaves (params.stats_*: in the launch up to 65,536 samples, behind it beyond) equals rato_risk_stats on that Z."""
    import torch
    from riskaversetrajopt_amd import stats
    d, n_u = _model(system, M, S)
    ws = stats.new_workspace(M, d.device)
    lib_flag = (d._lib.rato_drone_eval_stats_in_launch if system == "drone" else d._lib.rato_car_eval_stats_in_launch)(M)
    assert bool(lib_flag) == (M <= 65536)
    bufs = {}
    for k in range(4):
        us = _us(S, n_u, k)
        Z_ref, _, g_ref = d.eval_device(us, want_xs=True, want_g=True)          # plain kernel (trajectories wanted)
        Z, _, g = d.eval_device(us, want_g=True)                                # tiled kernel
        assert torch.equal(Z, Z_ref) and torch.equal(g, g_ref), k
        ref = stats.risk_stats_device(Z_ref, d.alpha)
        a = ref.cpu().numpy()
        for in_launch in (False, True):           # the statistics behind the kernel (default) / in its launch (stats_flags)
            Z2, rec = d.mc_step_device(us, workspace=ws, out=bufs, in_launch=in_launch)
            torch.cuda.synchronize()
            assert torch.equal(Z2, Z_ref)
            b = rec.cpu().numpy()
            assert np.array_equal(a[EXACT], b[EXACT]), (k, in_launch, a, b)
            np.testing.assert_allclose(b[SUMS], a[SUMS], rtol=1e-12, atol=1e-300)
            assert not _signal_words(ws).any()
    st = d.monte_carlo_statistics(_us(S, n_u, 3))
    assert st["var"] == b[0] and st["max"] == b[4] and st["frac_satisfied"] == b[2]
    Zh = np.sort(Z_ref.double().cpu().numpy())
    assert b[0] == Zh[M - int(np.floor(d.alpha * M)) - 1] and b[4] == Zh[-1]


@pytest.mark.parametrize("system,M,S,K", [("drone", 10000, 50, 7), ("drone", 12289, 20, 3), ("drone", 300, 70, 120),
                                          ("driving", 10000, 40, 7), ("driving", 20000, 17, 2), ("driving", 65, 90, 33)])
def test_batched_sequences_equal_the_single_calls(system, M, S, K):
    """rato_*_eval_batch: K control sequences on one resident batch in one call (the reference's Monte-Carlo report loops
    over 120: drone_risk.py:697-725) -- row k is the single-sequence call's Z and record."""
    import torch
    from riskaversetrajopt_amd import stats
    d, n_u = _model(system, M, S)
    us_b = np.stack([_us(S, n_u, k) for k in range(K)])
    Zb, rec = d.eval_batch_device(us_b)
    Zb2, none = d.eval_batch_device(us_b, want_stats=False)
    torch.cuda.synchronize()
    assert none is None and torch.equal(Zb, Zb2) and tuple(Zb.shape) == (K, M)
    for k in range(0, K, max(1, K // 7)):
        Z, _, _ = d.eval_device(us_b[k])
        r = stats.risk_stats_device(Z, d.alpha).cpu().numpy()
        assert torch.equal(Zb[k], Z), k
        b = rec[k].cpu().numpy()
        assert np.array_equal(r[EXACT], b[EXACT]), (k, r, b)
        np.testing.assert_allclose(b[SUMS], r[SUMS], rtol=1e-12, atol=1e-300)
    with pytest.raises(ValueError):
        d.eval_batch_device(us_b[:, :-1])


@pytest.mark.parametrize("system,M,S", [("drone", 10000, 50), ("driving", 10000, 40), ("drone", 50000, 50)])
def test_monte_carlo_step_replayed_from_a_hipgraph(system, M, S):
    """BASELINE C2 / C3 in the reference's own form (drone_risk.py:643-725, driving.py:618-740): one captured launch per
    Monte-Carlo step; prints the replayed step time."""
    import torch
    from riskaversetrajopt_amd import stats
    d, n_u = _model(system, M, S)
    us = torch.zeros((S, n_u), dtype=torch.float32, device=d.device)
    ws = stats.new_workspace(M, d.device)
    st = torch.empty(stats.N_STATS, dtype=torch.float64, device=d.device)
    bufs = {}
    d.mc_step_device(us, workspace=ws, stats_out=st, out=bufs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Z, _ = d.mc_step_device(us, workspace=ws, stats_out=st, out=bufs)
    for k in range(5):
        u = torch.as_tensor(_us(S, n_u, k), dtype=torch.float32, device=d.device)
        us.copy_(u)
        g.replay()
        torch.cuda.synchronize()
        ref = stats.risk_stats_device(Z, d.alpha).cpu().numpy()
        b = st.cpu().numpy()
        assert np.array_equal(ref[EXACT], b[EXACT]), (k, ref, b)
        np.testing.assert_allclose(b[SUMS], ref[SUMS], rtol=1e-12, atol=1e-300)
        Z_ref, _, _ = d.eval_device(u, want_xs=True)
        assert torch.equal(Z, Z_ref)
        assert not _signal_words(ws).any()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    print(f"{system} M={M} S={S} Monte-Carlo step (one launch): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per replayed step")
