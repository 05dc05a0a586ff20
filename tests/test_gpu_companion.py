"""GPU: companion statistics (rato_saa.h, "COMPANION statistics"): the exact selection started BESIDE the row-parallel
linearize kernel on a second stream -- it waits until the kernel has counted every tile's Z in -- against the same two
launches one behind the other: identical records bit for bit, eager and as a two-branch hipGraph, with the signal words
back at zero after every step."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _signal_words(ws):
    import torch
    from riskaversetrajopt_amd import stats
    off = stats.signal_ptr(ws) - ws.data_ptr()
    return ws[off:off + 32].view(torch.int32).cpu().numpy()


def _us(S, n_u, k):
    t = np.arange(S)[:, None]
    base = np.hstack([0.6 * np.cos(0.3 * t + 0.1 * k) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    return base[:, :n_u] * (1.0 - 0.03 * k)


@pytest.mark.parametrize("system,M,S", [("drone", 3000, 20), ("drone", 10000, 50), ("drone", 40000, 20), ("drone", 100003, 20),
                                        ("driving", 3000, 20), ("driving", 10000, 40), ("driving", 125001, 40)])
def test_companion_step_equals_the_two_launches(system, M, S):
    import torch
    from riskaversetrajopt_amd import drone_risk, drone_utils, driving, stats
    if system == "drone":
        dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=3)
        d, n_u = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M), 3
    else:
        dW, x0, ws_, wr = driving.sample_uncertain_parameters_device(M, S, seed=3)
        d, n_u = driving.Model.from_device(S, dW, x0, ws_, wr, 'saa', 0.05), 2
    ws_a, ws_b = stats.new_workspace(M, d.device), stats.new_workspace(M, d.device)
    comp = stats.Companion(d.device)
    ra = rb = None
    for k in range(6):
        us = _us(S, n_u, k)
        ra, sa = d.step_device(us, out=ra, workspace=ws_a)
        rb, sb = d.step_device(us, out=rb, workspace=ws_b, companion=comp)
        torch.cuda.synchronize()
        assert torch.equal(sa, sb), (k, sa, sb)
        assert torch.equal(ra["Z"], rb["Z"]) and torch.equal(ra["g_up"], rb["g_up"])
        if system == "drone":
            assert torch.equal(ra["sums"], rb["sums"])
        assert not _signal_words(ws_b).any()                      # every counter and flag lowered again
    st = sb.cpu().numpy()
    Zh = np.sort(rb["Z"].double().cpu().numpy())
    alpha = d.alpha
    assert st[0] == Zh[M - int(np.floor(alpha * M)) - 1]


@pytest.mark.parametrize("M", [10000, 60000])
def test_companion_as_a_second_branch_of_a_hipgraph(M):
    import torch
    from riskaversetrajopt_amd import drone_risk, drone_utils
    S = 50 if M == 10000 else 20
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=5)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
    plain = d.capture_step()
    comp = d.capture_step(companion=True)
    for k in range(8):
        us = _us(S, 3, k)
        plain.replay(us)
        comp.replay(us)
        torch.cuda.synchronize()
        assert torch.equal(plain.stats, comp.stats), k
        assert torch.equal(plain.out["sums"], comp.out["sums"])
        assert not _signal_words(comp.workspace).any()
    # replay time of the whole step (the statistics beside the kernel instead of behind it)
    for name, g in (("two launches in a row", plain), ("companion branch", comp)):
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        print(f"M={M} S={S} {name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per replayed step")


def test_companion_refuses_what_it_cannot_wait_for():
    import ctypes as C
    import torch
    from riskaversetrajopt_amd import _lib, drone_risk, drone_utils, stats
    M, S = 2000, 20
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=1)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
    ws = stats.new_workspace(M, d.device)
    with pytest.raises(_lib.RatoError):                              # the column kernel does not signal
        d.linearize_device(_us(S, 3, 0), cols_per_thread=8, samples_per_lane=1, factored=False,
                           signal=stats.signal_ptr(ws))
    Z = torch.zeros(2_000_000, device=d.device)
    with pytest.raises(_lib.RatoError):                              # beyond the one-launch forms of the selection
        stats.risk_stats_companion_device(Z, 0.1, workspace=stats.new_workspace(Z.numel(), d.device))
