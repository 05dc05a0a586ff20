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
