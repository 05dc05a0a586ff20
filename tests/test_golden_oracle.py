"""CPU: the oracle reproduces the committed golden fixtures (regression pin for
the checker itself) and the fixtures are self-consistent."""
import os

import numpy as np
import pytest

from oracle import drone as od, driving as ocar, hopper as oh, stats as ostats

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["drone_S20_M16", "drone_S50_M8"])
def test_drone_fixture(name):
    f = np.load(os.path.join(G, name + ".npz"))
    S, M = int(f["S"]), int(f["M"])
    rng = np.random.RandomState(0)
    DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
    assert np.array_equal(DWs, f["DWs"]) and np.array_equal(masses, f["masses"])
    model = od.Model(S, f["DWs"], f["masses"], f["obs_Qs"], 'saa', float(f["alpha"]))
    for kind in ("init", "graze"):
        us = f[f"{kind}_us"]
        np.testing.assert_allclose(model.us_to_state_trajectories(us), f[f"{kind}_xs"], rtol=1e-13, atol=1e-13)
        fdu, flo, _, gdu, gup = model.get_all_constraints_coeffs(us)
        np.testing.assert_allclose(gdu, f[f"{kind}_g_obs_du"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(gup, f[f"{kind}_g_up"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(fdu.mean(0), f[f"{kind}_final_du_mean"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(ostats.monte_carlo_avar(f[f"{kind}_Z"], 0.3), f[f"{kind}_avar"], rtol=1e-13)
    assert np.array_equal(f["init_us"][:, 2], np.zeros(S))       # z control of the initial guess is 0 (:119)


@pytest.mark.parametrize("name", ["driving_S20_M16", "driving_S40_M8"])
def test_driving_fixture(name):
    f = np.load(os.path.join(G, name + ".npz"))
    model = ocar.Model(f["states_init"], f["omegas_speed"], f["omegas_repulsive"], f["DWs"], 'saa', 0.05)
    for kind in ("init", "swerve"):
        us = f[f"{kind}_us"]
        np.testing.assert_allclose(model.us_to_state_trajectories(us), f[f"{kind}_xs"], rtol=1e-13, atol=1e-13)
        fdu, flo, _, gdu, gup = model.get_all_constraints_coeffs(us)
        np.testing.assert_allclose(gdu, f[f"{kind}_g_obs_du"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(gup, f[f"{kind}_g_up"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(fdu, f[f"{kind}_final_du"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name", ["hopper_S30_M30", "hopper_S60_M24"])
def test_hopper_fixture(name):
    f = np.load(os.path.join(G, name + ".npz"))
    model = oh.Model(f["intensities"], f["thetas"], f["taus"], 'saa', float(f["alpha"]), S=int(f["S"]))
    np.testing.assert_allclose(model.slip_risk_constraints(f["Z"]), f["gs"], rtol=1e-13, atol=1e-13)
    px, forces = model.contact_inputs(f["Z"])
    np.testing.assert_array_equal(px, f["px"])
    D1, D2 = model.slip_hessian_sums(px, forces, f["lam"])
    np.testing.assert_allclose(D1, f["D1"], rtol=1e-12)
    np.testing.assert_allclose(D2, f["D2"], rtol=1e-12)
