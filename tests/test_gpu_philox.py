"""GPU: the device-side sampler (csrc/philox.h, csrc/sampler.hip) against the NumPy restatement of
Philox4x32-10 (oracle/philox.py, pinned by Random123's known-answer vectors): the integer stream bit for bit, the
uniform / Box-Muller transforms to a stated fp32 tolerance, the three system samplers against the oracle's
restatement in the reference's layouts, distribution checks on 1e6 draws, and
eval(noise regenerated in the kernel) == eval(materialised noise) BITWISE.
Distributions sampled: drone_utils.py:61-93, driving.py:84-120, hopper.py:70-74."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats as sstats

pytestmark = pytest.mark.gpu

# Box-Muller on the hardware log2 / sqrt / sin / cos (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32; 1 ulp each, the
# trig on arguments in revolutions): measured max |device - fp64| on 4e6 normals is below 2e-6; stated bound:
NORMAL_ATOL = 4e-6


def _lib():
    from riskaversetrajopt_amd import _lib
    return _lib, _lib.load()


def test_integer_stream_is_bit_exact():
    import torch
    from oracle import philox as ph
    L, lib = _lib()
    T, M, ld, seed, sid = 5, 1000, 1004, 0x123456789ABCDEF, 3
    out = torch.zeros((T, 4, ld), dtype=torch.int32, device="cuda:0")
    assert lib.rato_philox_u32(L.ptr(out), T, M, ld, seed, sid, L.current_stream()) == 0
    got = out.cpu().numpy().view(np.uint32)
    t, m = np.meshgrid(np.arange(T), np.arange(M), indexing="ij")
    want = ph.philox_at(seed, ph.STREAM_USER + sid, t, m)
    for k in range(4):
        assert np.array_equal(got[:, k, :M], want[k])
    assert not got[:, :, M:].any()                                   # padding lanes are not written
    # bad arguments
    assert lib.rato_philox_u32(L.ptr(out), 0, M, ld, seed, sid, L.current_stream()) == -1
    assert lib.rato_philox_u32(L.ptr(out), T, M, M - 1, seed, sid, L.current_stream()) == -1


@pytest.mark.parametrize("C_", [1, 2, 3, 4])
def test_generic_fills_match_the_oracle_transforms(C_):
    import torch
    from oracle import philox as ph
    L, lib = _lib()
    T, M, seed = 7, 4099, 42
    out = torch.empty((T, C_, M), dtype=torch.float32, device="cuda:0")
    scale = (C.c_float * 4)(1.5, 0.5, 2.0, 1.0)
    mean = (C.c_float * 4)(0.0, 1.0, -2.0, 3.0)
    assert lib.rato_philox_normal(L.ptr(out), T, C_, M, M, seed, 0, scale, mean, L.current_stream()) == 0
    want = ph.normals(seed, ph.STREAM_USER, T, M, C_) * np.array(scale[:C_])[None, :, None] + np.array(mean[:C_])[None, :, None]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=NORMAL_ATOL * 2.0)
    assert lib.rato_philox_uniform(L.ptr(out), T, C_, M, M, seed, 1, scale, mean, L.current_stream()) == 0
    want = ph.uniforms(seed, ph.STREAM_USER + 1, T, M, C_) * np.array(scale[:C_])[None, :, None] + np.array(mean[:C_])[None, :, None]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=3e-7, atol=3e-7)
    assert lib.rato_philox_normal(L.ptr(out), T, 5, M, M, seed, 0, None, None, L.current_stream()) == -1


def test_system_samplers_match_the_oracle_restatement():
    import torch
    from oracle import philox as ph
    from riskaversetrajopt_amd import drone_utils, driving, hopper, drone_params, driving_params
    M, S, seed = 1001, 20, 99
    dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
    ld = mass.numel()
    DWs, masses, obs_Qs = ph.drone_sample(seed, ld, S, drone_params.T / S)
    np.testing.assert_allclose(dW.cpu().numpy(), np.transpose(DWs[:, :, 3:6], (1, 2, 0)), rtol=0,
                               atol=NORMAL_ATOL * np.sqrt(drone_params.T / S))
    np.testing.assert_allclose(mass.cpu().numpy(), masses, rtol=3e-7)
    np.testing.assert_allclose(Qsym[:, 0].cpu().numpy(), obs_Qs[:, :, 0, 0].T, rtol=2e-6)
    np.testing.assert_allclose(Qsym[:, 2].cpu().numpy(), obs_Qs[:, :, 1, 1].T, rtol=2e-6)
    assert not Qsym[:, 1].any()
    dWc, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=seed)
    x0_o, ws_o, wr_o, DWs_o = ph.car_sample(seed, M, S, driving_params.T / S, driving_params.state_init,
                                           [1e-1, 1e-1, 1e-4, 1e-4])
    np.testing.assert_allclose(dWc.cpu().numpy(), np.transpose(DWs_o[:, :, 6:8], (1, 2, 0)), rtol=0, atol=NORMAL_ATOL)
    np.testing.assert_allclose(x0.cpu().numpy(), x0_o[:, 4:].T, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ws.cpu().numpy(), ws_o, rtol=3e-7)
    np.testing.assert_allclose(wr.cpu().numpy(), wr_o, rtol=3e-7)
    a, th, tau = hopper.sample_friction_fields_device(M, seed=seed)
    a_o, th_o, tau_o = ph.hopper_sample(seed, M)
    np.testing.assert_allclose(a.cpu().numpy(), a_o.T, rtol=3e-7)
    np.testing.assert_allclose(th.cpu().numpy(), th_o.T, rtol=3e-7)
    np.testing.assert_allclose(tau.cpu().numpy(), tau_o.T, rtol=3e-7)
    # different seeds / ranks give different batches; the same seed reproduces bit for bit
    dW2, _, _ = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
    dW3, _, _ = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed + 1)
    assert torch.equal(dW, dW2) and not torch.equal(dW, dW3)


def test_distributions_on_1e6_draws():
    """moments and Kolmogorov-Smirnov distance of 1e6 device draws per distribution"""
    import torch
    L, lib = _lib()
    M = 250000
    out = torch.empty((1, 4, M), dtype=torch.float32, device="cuda:0")
    assert lib.rato_philox_normal(L.ptr(out), 1, 4, M, M, 2024, 5, None, None, L.current_stream()) == 0
    z = out.cpu().numpy().astype(np.float64).ravel()                 # 1e6 normals
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1.0) < 3e-3
    assert abs(sstats.skew(z)) < 1e-2 and abs(sstats.kurtosis(z)) < 2e-2
    assert sstats.kstest(z, "norm").statistic < 2.5e-3               # 1.63 / sqrt(1e6) = 1.6e-3 at the 1 % level
    c = np.corrcoef(out.cpu().numpy()[0].astype(np.float64))         # the 4 components are uncorrelated
    assert np.abs(c - np.eye(4)).max() < 8e-3
    assert lib.rato_philox_uniform(L.ptr(out), 1, 4, M, M, 2024, 6, None, None, L.current_stream()) == 0
    u = out.cpu().numpy().astype(np.float64).ravel()
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 1e-3 and abs(u.var() - 1 / 12) < 5e-4
    assert sstats.kstest(u, "uniform").statistic < 2.5e-3
    # lag-1 autocorrelation along the sample index and along the step index
    seq = torch.empty((64, 1, 16384), dtype=torch.float32, device="cuda:0")
    assert lib.rato_philox_normal(L.ptr(seq), 64, 1, 16384, 16384, 7, 7, None, None, L.current_stream()) == 0
    s = seq.cpu().numpy()[:, 0].astype(np.float64)
    assert abs(np.mean(s[:, 1:] * s[:, :-1])) < 5e-3 and abs(np.mean(s[1:] * s[:-1])) < 5e-3


def test_eval_with_regenerated_noise_is_bitwise_the_materialised_eval():
    import torch
    from riskaversetrajopt_amd import drone_risk, drone_utils, driving
    S, M, seed = 50, 10007, 11
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
    a = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
    b = drone_risk.Model.from_device(S, None, mass, Qsym, 'saa', 0.1, M=M, noise_seed=seed)
    Za, xa, ga = a.eval_device(us, want_xs=True, want_g=True)
    Zb, xb, gb = b.eval_device(us, want_xs=True, want_g=True)
    assert torch.equal(Za, Zb) and torch.equal(xa, xb) and torch.equal(ga, gb)
    sa, sb = a.monte_carlo_statistics(us), b.monte_carlo_statistics(us)
    assert sa == sb
    # the linearization regenerates its noise tile by tile (rato_drone_linearize_philox): bitwise the materialised one,
    # both output representations, with the step-Jacobian table
    from riskaversetrajopt_amd.drone_risk import untile
    for fact in (True, False):
        ra = a.linearize_device(us, factored=fact, want_A22=fact)
        rb = b.linearize_device(us, factored=fact, want_A22=fact)
        assert torch.equal(untile(ra["G"], M), untile(rb["G"], M))
        for k in ("g_up", "Z", "sums", "part") + (("W", "A22") if fact else ()):
            assert torch.equal(ra[k], rb[k]), (fact, k)
    with pytest.raises(Exception):
        b.linearize_device(us, cols_per_thread=8)                    # the column kernels read a materialised dW: loud
    S, M = 40, 8191
    us = np.hstack([0.4 * np.cos(0.4 * np.arange(S)[:, None]) - 0.2, 0.05 * np.sin(0.35 * np.arange(S)[:, None]) + 0.01]) * 0.5
    dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=seed)
    a = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)
    b = driving.Model.from_device(S, None, x0, ws, wr, 'saa', 0.05, noise_seed=seed)
    Za, xa, ga = a.eval_device(us, want_xs=True, want_g=True)
    Zb, xb, gb = b.eval_device(us, want_xs=True, want_g=True)
    assert torch.equal(Za, Zb) and torch.equal(xa, xb) and torch.equal(ga, gb)
    ra, rb = a.linearize_device(us), b.linearize_device(us)          # rato_car_linearize_philox: one-tile launches
    assert torch.equal(untile(ra["G"], M), untile(rb["G"], M))
    for k in ("g_up", "Z", "final_du", "final_rhs"):
        assert torch.equal(ra[k], rb[k]), k
    M = 70001                                                        # more tiles than workgroup slots: the tile queue
    dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=seed + 1)
    a = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)
    b = driving.Model.from_device(S, None, x0, ws, wr, 'saa', 0.05, noise_seed=seed + 1)
    ra, rb = a.linearize_device(us), b.linearize_device(us)
    assert torch.equal(untile(ra["G"], M), untile(rb["G"], M)) and torch.equal(ra["g_up"], rb["g_up"])
    assert torch.equal(ra["Z"], rb["Z"])
    with pytest.raises(Exception):
        b.linearize_device(us, cols_per_thread=8)


def test_philox_batch_through_the_oracle_rollout():
    """the device batch, read back, rolled out by the fp64 oracle == the device rollout (fp32 tolerance)"""
    from oracle import drone as od, philox as ph
    from riskaversetrajopt_amd import drone_risk, drone_utils, drone_params
    from tests import _tol as tol
    S, M, seed = 20, 512, 5
    dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
    d = drone_risk.Model.from_device(S, None, mass, Qsym, 'saa', 0.1, M=M, noise_seed=seed)
    DWs, masses, obs_Qs = ph.drone_sample(seed, M, S, drone_params.T / S)
    o = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
    us = o.initial_guess_us_mat() + 0.3
    np.testing.assert_allclose(d.us_to_state_trajectories(us), o.us_to_state_trajectories(us),
                               rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)
