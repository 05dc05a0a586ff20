"""Pins the driving oracle: independent autodiff, finite differences and the
structural invariants of driving.py:145-298 (parity is unpinned by the reference)."""
import numpy as np
import pytest
import torch
from torch.func import jacfwd, vmap

from oracle import driving as ocar
from tests import _torch_forward as tf


def make_model(M=6, S=20, seed=0, method='saa'):
    rng = np.random.RandomState(seed)
    return ocar.Model(*ocar.sample_uncertain_parameters(rng, M, method, S), method=method)


def swerving_us(S):
    t = np.arange(S)[:, None]
    return np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01])


def test_sampler_matches_reference_loop_order():
    M, S = 3, 20
    rng = np.random.RandomState(0)
    x0, ws, wr, DWs = ocar.sample_uncertain_parameters(rng, M, 'saa', S)
    np.random.seed(0)
    ws_ref = np.random.uniform(0.1 - 0.075, 0.1 + 0.075, M)
    wr_ref = np.random.uniform(0.05 - 0.045, 0.05 + 0.045, M)
    x0_ref = np.repeat(ocar.state_init[None], M, axis=0)
    for i in range(M):
        x0_ref[i, 4:] = x0_ref[i, 4:] + np.diag([1e-1, 1e-1, 1e-4, 1e-4]) @ np.random.randn(4)
    DW_ref = np.zeros((M, S, 8))
    for i in range(M):
        for t in range(S):
            DW_ref[i, t, :] = np.random.randn(8)
    DW_ref = np.sqrt(0.5) * DW_ref
    np.testing.assert_allclose(ws, ws_ref, rtol=0, atol=0)
    np.testing.assert_allclose(wr, wr_ref, rtol=0, atol=0)
    np.testing.assert_allclose(x0, x0_ref, rtol=1e-15, atol=0)
    np.testing.assert_array_equal(DWs, DW_ref)


@pytest.mark.parametrize("S,us_kind", [(20, 'init'), (20, 'swerve'), (40, 'swerve')])
def test_linearization_matches_autodiff(S, us_kind):
    model = make_model(M=5, S=S)
    us = model.initial_guess_us_mat() if us_kind == 'init' else swerving_us(S)
    fdu, flo, fup, gdu, gup = model.get_all_constraints_coeffs(us)

    def fwd(u, x0, ws, wr, dWs):
        return tf.driving_forward(u, x0, ws, wr, dWs, S, model.dt, float(ocar.min_separation_distance))

    args = (torch.tensor(us), torch.tensor(model.states_init), torch.tensor(model.omegas_speed),
            torch.tensor(model.omegas_repulsive), torch.tensor(model.DWs))
    dims = (None, 0, 0, 0, 0)
    vals = vmap(fwd, in_dims=dims)(*args)
    jac = vmap(jacfwd(fwd), in_dims=dims)(*args)
    M = model.M
    v_final_du = jac[0].reshape(M, 4, 2 * S).numpy()
    g_obs_du = jac[1].reshape(M, S, 2 * S).numpy()
    xs = model.us_to_state_trajectories(us)
    np.testing.assert_allclose(model.final_constraints(xs), vals[0].numpy(), rtol=0, atol=1e-11)
    np.testing.assert_allclose(-model.separation_distances_at_all_times(xs), vals[1].numpy(),
                               rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fdu, v_final_du, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gdu, g_obs_du, rtol=1e-9, atol=1e-11)
    uvec = us.reshape(-1)
    np.testing.assert_allclose(flo, -vals[0].numpy() + v_final_du @ uvec, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gup, -vals[1].numpy() + g_obs_du @ uvec, rtol=1e-9, atol=1e-10)


def test_linearization_matches_finite_differences():
    S = 20
    model = make_model(M=3, S=S)
    us = swerving_us(S)
    fdu, _, _, gdu, _ = model.get_all_constraints_coeffs(us)
    eps = 1e-6
    for (s, i) in [(0, 0), (0, 1), (9, 1), (18, 0)]:
        up, um = us.copy(), us.copy()
        up[s, i] += eps
        um[s, i] -= eps
        xp, xm = model.us_to_state_trajectories(up), model.us_to_state_trajectories(um)
        dg = -(model.separation_distances_at_all_times(xp)
               - model.separation_distances_at_all_times(xm)) / (2 * eps)
        df = (model.final_constraints(xp) - model.final_constraints(xm)) / (2 * eps)
        np.testing.assert_allclose(gdu[..., s * 2 + i], dg, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(fdu[..., s * 2 + i], df, rtol=1e-5, atol=1e-7)


def test_structural_invariants():
    S = 20
    model = make_model(M=4, S=S)
    us = swerving_us(S)
    xs = model.us_to_state_trajectories(us)
    # the ego sub-state is bit-identical across samples
    assert np.all(xs[:, :, :4] == xs[0:1, :, :4])
    fdu, flo, _, gdu, _ = model.get_all_constraints_coeffs(us)
    assert np.all(fdu == fdu[0:1]) and np.all(flo == flo[0:1])
    for t in range(S):
        for s in range(S):
            if s > t - 1:
                assert np.all(gdu[:, t, 2 * s:2 * s + 2] == 0.0)
    assert np.count_nonzero(gdu[0]) == S * (S - 1)
    assert np.count_nonzero(fdu[0]) == 116   # SURVEY.md facts table (S=20)


def test_baseline_zeroes_gains_and_noise():
    model = make_model(M=4, S=20, method='baseline')
    assert np.all(model.DWs == 0) and np.all(model.omegas_speed == 0) and np.all(model.omegas_repulsive == 0)
    # baseline does NOT draw the randn(4) block (driving.py:105)
    assert np.all(model.states_init == ocar.state_init[None])
    xs = model.us_to_state_trajectories(swerving_us(20))
    # pedestrian walks straight at constant velocity
    np.testing.assert_allclose(xs[:, -1, 5], -6.0 + 1.3 * ocar.T, rtol=1e-12)


def test_dense_qp_rows_layout():
    S, M = 20, 3
    model = make_model(M=M, S=S)
    us = swerving_us(S)
    A, low, up = model.get_all_constraints_coeffs_all(us)
    assert A.shape == (4 + 1 + M + M * S + 1, 2 * S + M + 2)
    fdu, flo, fup, gdu, gup = model.get_all_constraints_coeffs(us)
    i, t = 1, 11
    row = 4 + 1 + M + i * S + t
    np.testing.assert_array_equal(A[row, :2 * S], gdu[i, t])
    assert A[row, 2 * S + i] == -1.0 and A[row, -1] == -1.0 and up[row] == gup[i, t]
