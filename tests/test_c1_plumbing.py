"""BASELINE config C1 — "drone_gaussian.py linear dynamics + Gaussian wind, M=100 T=30, CPU NumPy
reference (plumbing, no GPU)".  The named script has no sample axis; as SURVEY.md §8(d) reads it: run the CPU
oracle end to end at that size (sample -> rollout -> linearize -> reduce -> assemble -> solve one QP) and check
the script's mean/covariance recursion (drone_gaussian.py:161-227) against the sample moments of the rollouts."""
import numpy as np

from oracle import drone as od, gaussian as og, stats as ostats
from tests._oracle_qp import DroneOracleQP

S, M = 30, 100


def _us():
    t = np.arange(S)[:, None]
    return np.hstack([0.25 * np.cos(0.2 * t) + 0.1, 0.05 * np.sin(0.3 * t), 0.02 * np.cos(t)])


def test_end_to_end_cpu_plumbing():
    DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(0), 'saa', M=M, S=S)
    o = od.Model(S, DWs, masses, Q, 'saa', 0.1)
    us = _us()
    xs = o.us_to_state_trajectories(us)
    assert xs.shape == (M, S + 1, 6)
    qp_model = DroneOracleQP(o)
    A, l, u = qp_model.get_constraints_coeffs(us, 2)
    assert A.shape == (6 + 1 + M + M * 3 * S + 1 + 3 * S, 3 * S + M + 2)
    assert A.nnz == np.count_nonzero(A.toarray())
    qp_model.define_problem(us)
    us_new, t_risk = qp_model.solve()
    assert us_new.shape == (S, 3) and np.all(np.abs(us_new) <= od.u_max + 1e-6) and np.isfinite(t_risk)
    ok, Z = o.monte_carlo_no_collisions_constraint_verification(us_new)
    assert ostats.monte_carlo_avar(Z, 0.1) >= ostats.monte_carlo_var(Z, 0.1)


def test_gaussian_recursion_matches_sample_moments():
    rng = np.random.RandomState(1)
    Mbig = 4000                                  # tighter statistics than M=100 for the comparison
    # unit-variance increments: the SAA rollout then injects dt (beta/m)^2 per step, the Gaussian Sigma_w
    DWs, _, Q = od.sample_uncertain_parameters(rng, 'saa', M=Mbig, S=S, dt=1.0)
    masses = np.full(Mbig, od.mass_nom)          # Brownian noise only (the recursion treats the mass as iid per step)
    o = od.Model(S, DWs, masses, Q)
    us = _us()
    xs = o.us_to_state_trajectories(us)
    mu = og.mean_trajectory(us, S)
    Sig = og.covariance_trajectory(us, S, mass_variance=0.0)
    np.testing.assert_allclose(xs.mean(0), mu, atol=5e-3)
    emp = np.cov(xs[:, -1, :].T)
    np.testing.assert_allclose(np.diag(emp), np.diag(Sig[-1]), rtol=0.15)
    # C1-sized batch: same check at M = 100 with the looser statistical tolerance
    emp100 = np.cov(xs[:M, -1, :].T)
    assert np.all(np.diag(emp100) < 2.0 * np.diag(Sig[-1])) and np.all(np.diag(emp100) > 0.5 * np.diag(Sig[-1]))
    # with the mass term the propagated variance can only grow (rank-one form and the reference's scalar form alike)
    for outer in (True, False):
        Sig_m = og.covariance_trajectory(us, S, outer_product=outer)
        assert np.all(np.diag(Sig_m[-1]) >= np.diag(Sig[-1]) - 1e-15)
