"""CPU: the oracle's C restatement agrees with its NumPy restatement (both are test infrastructure)."""
import numpy as np
import pytest

from oracle import c_oracle, drone as od


@pytest.mark.parametrize("S,M", [(20, 9), (50, 5)])
def test_c_oracle_matches_numpy_oracle(S, M):
    DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(0), 'saa', M=M, S=S)
    o = od.Model(S, DWs, masses, Q)
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    c = c_oracle.drone(us, DWs, masses, Q, o.dt, nthreads=2)
    fdu, flo, _, gdu, gup = o.get_all_constraints_coeffs(us)
    np.testing.assert_allclose(c["xs"], o.us_to_state_trajectories(us), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(c["v_final_du"], fdu, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(c["val_final"], flo, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(c["g_obs_du"], gdu, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(c["g_up"], gup, rtol=1e-10, atol=1e-11)
    _, Z = o.monte_carlo_no_collisions_constraint_verification(us)
    np.testing.assert_allclose(c["Z"], Z, rtol=1e-12, atol=1e-13)
    assert np.array_equal(c["g_obs_du"] == 0, gdu == 0)
    one = c_oracle.drone(us, DWs, masses, Q, o.dt, nthreads=1)
    assert np.array_equal(one["g_obs_du"], c["g_obs_du"])       # thread count does not change results


def test_c_oracle_streaming_form_equals_the_dense_form():
    """rato_oracle_drone_stream (bench.py's CPU baseline at M = 1e5: dense rows formed per thread, reduced on the fly)"""
    from oracle import c_oracle, drone as od
    S, M = 20, 300
    DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(3), 'saa', M=M, S=S)
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)])
    dense = c_oracle.drone(us, DWs, masses, Q, od.T / S, nthreads=2)
    for nt in (1, 3):
        st = c_oracle.drone_stream(us, DWs, masses, Q, od.T / S, nthreads=nt)
        np.testing.assert_allclose(st["sum_final_du"], dense["v_final_du"].sum(0), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(st["sum_val_final"], dense["val_final"].sum(0), rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(st["Z"], dense["Z"])
        np.testing.assert_allclose(st["checksum"][0], dense["g_obs_du"].sum() + dense["g_up"].sum(), rtol=1e-11)


@pytest.mark.parametrize("S,M,method", [(20, 9, 'saa'), (40, 5, 'saa'), (20, 4, 'baseline')])
def test_c_car_oracle_matches_numpy_oracle(S, M, method):
    """car_sample (oracle/saa_oracle.c) == oracle/driving.py (which is pinned by executing the reference's text)"""
    from oracle import driving as ocar
    samples = ocar.sample_uncertain_parameters(np.random.RandomState(2), M, method, S)
    o = ocar.Model(*samples, method=method, alpha=0.1)
    t = np.arange(S)[:, None]
    us = np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01]) * (20.0 / S)
    c = c_oracle.car(us, *samples, nthreads=2)
    _, _, _, gdu, gup = o.get_all_constraints_coeffs(us)
    np.testing.assert_allclose(c["xs"], o.us_to_state_trajectories(us), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(c["g_obs_du"], gdu, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(c["g_up"], gup, rtol=1e-10, atol=1e-11)
    _, Z = o.monte_carlo_separation_constraints_verification(us)
    np.testing.assert_allclose(c["Z"], Z, rtol=1e-12, atol=1e-13)
    assert np.array_equal(c["g_obs_du"] == 0, gdu == 0)


@pytest.mark.parametrize("system", ["drone", "driving"])
def test_streaming_cut_oracle_equals_dense_rows(system):
    """rato_oracle_*_rowmax / _tail_rows (one sample's linearization at a time) == the same sums over the dense rows"""
    rng = np.random.RandomState(5)
    if system == "drone":
        S, M = 20, 41
        DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(1), 'saa', M=M, S=S)
        o = od.Model(S, DWs, masses, Q)
        co = c_oracle.DroneCutOracle(DWs, masses, Q, o.dt, nthreads=3)
        t = np.arange(S)[:, None]
        uk = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)])
        fdu, flo, _, gdu, gup = o.get_all_constraints_coeffs(uk)
        fd, fl = co.final_rows(uk)
        np.testing.assert_allclose(fd, fdu.mean(0), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(fl, flo.mean(0), rtol=1e-11, atol=1e-12)
    else:
        from oracle import driving as ocar
        S, M = 20, 33
        samples = ocar.sample_uncertain_parameters(np.random.RandomState(1), M, 'saa', S)
        o = ocar.Model(*samples)
        co = c_oracle.CarCutOracle(*samples, nthreads=3)
        t = np.arange(S)[:, None]
        uk = np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01])
        _, _, _, gdu, gup = o.get_all_constraints_coeffs(uk)
    nU = uk.size
    G, gup = gdu.reshape(M, -1, nU), gup.reshape(M, -1)
    u = uk.reshape(-1) + 0.2 * rng.randn(nU)
    rows = G @ u - gup
    m, arg = co.rowmax(uk, u)
    np.testing.assert_array_equal(arg, rows.argmax(1))
    np.testing.assert_allclose(m, rows.max(1), rtol=1e-12, atol=1e-12)
    K = 3
    w = rng.rand(K, M) * (rng.rand(K, M) < 0.4)
    args = rng.randint(0, rows.shape[1], size=(K, M)).astype(np.int32)
    grad, gs = co.tail_rows(uk, w, args)
    idx = np.arange(M)
    for k in range(K):
        np.testing.assert_allclose(grad[k], (w[k][:, None] * G[idx, args[k]]).sum(0), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(gs[k], w[k] @ gup[idx, args[k]], rtol=1e-12, atol=1e-13)
