"""Pins the drone oracle (parity is unpinned by the reference — no tests or
golden vectors exist there): independent autodiff, finite differences and the
structural invariants of drone_risk.py:122-280."""
import numpy as np
import pytest
import torch
from torch.func import jacfwd, vmap

from oracle import drone as od
from tests import _torch_forward as tf


def make_model(M=6, S=20, seed=0, method='saa', dt_sampler=od.DT_MODULE):
    rng = np.random.RandomState(seed)
    DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, method, M=M, S=S, dt=dt_sampler)
    return od.Model(S, DWs, masses, obs_Qs, method, 0.1)


def grazing_us(S):
    """A hand-made iterate that flies towards the obstacles (non-trivial g, |v| != 0)."""
    t = np.arange(S)[:, None]
    return np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)])


@pytest.mark.parametrize("S", [20, 50])
def test_sampler_matches_reference_loop_order(S):
    # the reference draws randn(6) per (i,t) in nested loops (drone_utils.py:88-90)
    M = 4
    rng = np.random.RandomState(0)
    DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
    np.random.seed(0)
    masses_ref = np.random.uniform(29, 35, M)
    radii = np.zeros((3, 3, M))
    for j in range(3):
        for d in range(3):
            radii[j, d] = np.random.uniform(-0.025, 0.025, M)
    DW_ref = np.zeros((M, S, 6))
    for i in range(M):
        for t in range(S):
            DW_ref[i, t, :] = np.sqrt(2.5) * np.random.randn(6)
    assert np.array_equal(masses, masses_ref)
    assert np.array_equal(DWs, DW_ref)
    for j in range(3):
        for d in range(3):
            assert np.array_equal(obs_Qs[:, j, d, d], 1.0 / (od.obs_radii[j] + radii[j, d])**2)


@pytest.mark.parametrize("S,us_kind", [(20, 'init'), (20, 'graze'), (50, 'graze')])
def test_linearization_matches_autodiff(S, us_kind):
    model = make_model(M=5, S=S)
    us = model.initial_guess_us_mat() if us_kind == 'init' else grazing_us(S)
    fdu, flo, fup, gdu, gup = model.get_all_constraints_coeffs(us)
    us_t = torch.tensor(us)

    def fwd(u, mass, dWs, Q):
        return tf.drone_forward(u, mass, dWs, Q, S, model.dt)

    vals = vmap(fwd, in_dims=(None, 0, 0, 0))(
        us_t, torch.tensor(model.masses), torch.tensor(model.DWs), torch.tensor(model.obs_Qs))
    jac = vmap(jacfwd(fwd), in_dims=(None, 0, 0, 0))(
        us_t, torch.tensor(model.masses), torch.tensor(model.DWs), torch.tensor(model.obs_Qs))
    M = model.M
    v_final_du = jac[0].reshape(M, 6, 3 * S).numpy()
    g_obs_du = jac[1].reshape(M, 3, S, 3 * S).numpy()
    xs = model.us_to_state_trajectories(us)
    np.testing.assert_allclose(model.final_constraints(xs), vals[0].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(model.obstacle_avoidance_constraints(xs, model.obs_Qs),
                               vals[1].numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fdu, v_final_du, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gdu, g_obs_du, rtol=1e-10, atol=1e-12)
    uvec = us.reshape(-1)
    np.testing.assert_allclose(flo, -vals[0].numpy() + v_final_du @ uvec, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(gup, -vals[1].numpy() + g_obs_du @ uvec, rtol=1e-10, atol=1e-11)
    assert np.array_equal(flo, fup)


def test_linearization_matches_finite_differences():
    S = 20
    model = make_model(M=3, S=S)
    us = grazing_us(S)
    _, _, _, gdu, _ = model.get_all_constraints_coeffs(us)
    fdu = model.get_all_constraints_coeffs(us)[0]
    eps = 1e-6
    for (s, a) in [(0, 0), (3, 1), (10, 0), (18, 1), (7, 2)]:
        up, um = us.copy(), us.copy()
        up[s, a] += eps
        um[s, a] -= eps
        xp, xm = model.us_to_state_trajectories(up), model.us_to_state_trajectories(um)
        dg = (model.obstacle_avoidance_constraints(xp, model.obs_Qs)
              - model.obstacle_avoidance_constraints(xm, model.obs_Qs)) / (2 * eps)
        df = (model.final_constraints(xp) - model.final_constraints(xm)) / (2 * eps)
        np.testing.assert_allclose(gdu[..., s * 3 + a], dg, rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(fdu[..., s * 3 + a], df, rtol=2e-6, atol=2e-8)


def test_structural_invariants():
    S = 20
    model = make_model(M=4, S=S)
    us = grazing_us(S)
    fdu, _, _, gdu, _ = model.get_all_constraints_coeffs(us)
    # causality: d g_t / d u_s == 0 exactly unless s <= t-1; z-control column is exactly 0
    for t in range(S):
        for s in range(S):
            blk = gdu[:, :, t, s * 3:(s + 1) * 3]
            if s > t - 1:
                assert np.all(blk == 0.0)
            assert np.all(blk[..., 2] == 0.0)
    assert np.all(gdu[:, :, 0, :] == 0.0)
    nnz = np.count_nonzero(gdu[0])
    assert nnz == od.n_obs * S * (S - 1)
    # axes decouple in the final-state Jacobian
    for a in range(3):
        for b in range(3):
            if a != b:
                assert np.all(fdu[:, a, b::3] == 0.0)
                assert np.all(fdu[:, 3 + a, b::3] == 0.0)


def test_baseline_is_zero_noise_nominal_special_case():
    S, M = 20, 5
    base = make_model(M=M, S=S, method='baseline')
    assert np.all(base.DWs == 0) and np.all(base.masses == od.mass_nom)
    us = grazing_us(S)
    xs = base.us_to_state_trajectories(us)
    assert np.all(xs == xs[0:1])            # every "sample" is the same nominal rollout
    saa = make_model(M=M, S=S)
    nom = od.Model(S, 0 * saa.DWs, np.full(M, od.mass_nom), base.obs_Qs, 'saa')
    np.testing.assert_array_equal(nom.us_to_state_trajectories(us), xs)


def test_dense_qp_rows_layout():
    S, M = 20, 3
    model = make_model(M=M, S=S)
    us = grazing_us(S)
    A, low, up = model.get_all_constraints_coeffs_all(us)
    R_s = od.n_obs * S
    assert A.shape == (6 + 1 + M + M * R_s + 1, 3 * S + M + 2)
    fdu, flo, fup, gdu, gup = model.get_all_constraints_coeffs(us)
    np.testing.assert_array_equal(A[:6, :3 * S], fdu.mean(0))
    np.testing.assert_array_equal(low[:6], up[:6])
    assert A[6, -1] == M * model.alpha and np.all(A[6, 3 * S:-1] == 1.0)
    i, j, t = 2, 1, 7
    row = 6 + 1 + M + i * R_s + j * S + t
    np.testing.assert_array_equal(A[row, :3 * S], 0.01 * gdu[i, j, t])
    assert A[row, 3 * S + i] == -0.01 and A[row, -1] == -0.01 and up[row] == 0.01 * gup[i, j, t]
    assert low[row] == -np.inf and A[-1, -2] == -1.0


def test_monte_carlo_and_cvar_identities():
    from oracle import stats
    S, M = 20, 400
    model = make_model(M=M, S=S, seed=3)
    us = grazing_us(S)
    ok, Z = model.monte_carlo_no_collisions_constraint_verification(us)
    assert ok.dtype == bool and Z.shape == (M,)
    for alpha in (0.05, 0.1, 0.3):
        var = stats.monte_carlo_var(Z, alpha)
        cvar = stats.monte_carlo_avar(Z, alpha)
        assert cvar >= var
        k = int(np.floor(alpha * M))
        assert np.sum(Z > var) <= k < np.sum(Z >= var) + 0
        # VaR minimises the Rockafellar-Uryasev function
        for t in np.linspace(Z.min(), Z.max(), 41):
            assert stats.rockafellar_uryasev(Z, alpha, t) >= cvar - 1e-12
        # alpha*M integer -> CVaR = mean of the k largest
        if abs(alpha * M - k) < 1e-9:
            np.testing.assert_allclose(cvar, np.sort(Z)[-k:].mean(), rtol=1e-12)
