"""CPU: the reduced (u, slack) formulation and its cutting-plane loop (riskaversetrajopt_amd.cvar_cuts) against the
reference's full QP, everything in fp64 on the oracle's linearization (tests/_host_cuts.py supplies the cut oracle in
NumPy).  What the device path adds on top of this is only the fp32 linearization data and where the sums are formed;
the algebra -- elimination of y / t, the 'baseline' rows as a one-sample tail, the relaxed first iterations
(drone_risk.py:413-417, driving.py:411-415), recycled cuts, the delta form of the rows -- is checked here."""
import numpy as np
import pytest

from oracle import drone as od, driving as ocar
from riskaversetrajopt_amd import scp
from tests._host_cuts import DroneReducedOracle, DrivingReducedOracle
from tests._oracle_qp import DroneOracleQP, DrivingOracleQP


def _drone(M, S, alpha, method, seed=0):
    DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(seed), method, M=M, S=S)
    return od.Model(S, DWs, masses, Q, method, alpha)


def _car(M, S, alpha, method, seed=0):
    samples = ocar.sample_uncertain_parameters(np.random.RandomState(seed), M, method, S)
    return ocar.Model(*samples, method=method, alpha=alpha)


@pytest.mark.parametrize("method,M", [("saa", 30), ("baseline", 8)])
def test_drone_reduced_scp_equals_full_qp_scp(method, M):
    o = _drone(M, 20, 0.2, method)
    full = scp.run_drone(DroneOracleQP(o), num_scp_iters_max=12, warmup_iters=0)
    red = scp.run_drone_reduced(DroneReducedOracle(o), num_scp_iters_max=12)
    # measured: 1.1e-7 (saa; the polished full QP keeps delta = 1e-6 of regularisation), 8e-14 (baseline)
    np.testing.assert_allclose(red["us"], full["us"], rtol=0, atol=2e-6)
    assert abs(red["t_risk"] - full["t_risk"]) < 2e-6
    assert red["cuts"][:2].sum() == 0 and red["cuts"][2] >= 1          # relaxed iterations: no CVaR rows
    # the reference's expression G u - g_up and the delta form g + G (u - u_k) are the same rows
    ref_form = scp.run_drone_reduced(DroneReducedOracle(o, delta=False), num_scp_iters_max=12)
    np.testing.assert_allclose(ref_form["us"], red["us"], rtol=0, atol=1e-11)


@pytest.mark.parametrize("method,M", [("saa", 16), ("baseline", 8)])
def test_driving_reduced_scp_equals_full_qp_scp(method, M):
    o = _car(M, 20, 0.1, method)
    full = scp.run_driving(DrivingOracleQP(o), num_scp_iters_max=8)
    red = scp.run_driving_reduced(DrivingReducedOracle(o), num_scp_iters_max=8)
    np.testing.assert_allclose(red["us"], full["us"], rtol=0, atol=1e-8)            # measured 2e-11 / 5e-16
    assert abs(red["t_risk"] - full["t_risk"]) < 1e-8
    assert red["cuts"][0] == 0


def test_relaxed_iterations_leave_slack_at_minus_one():
    """scp_iter < 2: rows [n_x:] scaled by 1e-7 inside [-0.1, 0.1] (drone_risk.py:413-417) -- the slack row with them"""
    o = _drone(12, 20, 0.2, "saa")
    q = DroneOracleQP(o)
    us0 = o.initial_guess_us_mat()
    q.define_problem(us0)
    q.update_problem(us0, 0)
    us_full, _ = q.solve()
    us_red, t_red, info = DroneReducedOracle(o).solve_reduced(us0, 0)
    assert abs(q.res.x[-2] + 1.0) < 1e-6 and abs(info["slack"] + 1.0) < 1e-12 and t_red == 0.0
    np.testing.assert_allclose(us_red, us_full, rtol=0, atol=1e-9)


def test_recycled_cuts_do_not_change_the_iterates():
    o = _drone(60, 20, 0.1, "saa", seed=3)
    a = scp.run_drone_reduced(DroneReducedOracle(o), num_scp_iters_max=10)
    m = DroneReducedOracle(o)
    m.cs.recycle = False
    b = scp.run_drone_reduced(m, num_scp_iters_max=10)
    np.testing.assert_allclose(a["us"], b["us"], rtol=0, atol=1e-8)
    assert a["cuts"].sum() <= b["cuts"].sum()


@pytest.mark.parametrize("system,method,M,alpha", [("drone", "saa", 200, 0.1), ("drone", "saa", 1000, 0.05),
                                                   ("drone", "baseline", 40, 0.1), ("driving", "saa", 200, 0.1),
                                                   ("driving", "saa", 1000, 0.05), ("driving", "baseline", 40, 0.1)])
def test_reduced_solution_satisfies_the_kkt_conditions_of_the_full_qp(system, method, M, alpha):
    """At M = 200 .. 1000 the host ADMM + polish no longer identifies the active set of the (massively degenerate)
    full QP reliably, so instead of comparing with its output the reduced solution is CERTIFIED against the full QP
    itself: lifted to (u, y, slack, t) and with the master's multipliers spread over the rows of the reference's
    layout, it satisfies primal feasibility, stationarity, dual signs and complementarity of
    `l <= A z <= u`, `1/2 z'Pz + q'z` to 1e-9 -- it IS the optimum of the reference's QP (unique in u)."""
    from tests._host_cuts import kkt_certificate
    S = 20
    if system == "drone":
        o = _drone(M, S, alpha, method)
        fq, ro, n_c, n_u, R, kappa, first = DroneOracleQP(o), DroneReducedOracle(o), 6, 3, 3, 0.01, 2
    else:
        o = _car(M, S, alpha, method)
        fq, ro, n_c, n_u, R, kappa, first = DrivingOracleQP(o), DrivingReducedOracle(o), 4, 2, 1, 1.0, 1
    P, q = fq.get_objective_coeffs()
    us = o.initial_guess_us_mat()
    checked = 0
    for it in range(6):
        nxt, _, info = ro.solve_reduced(us, it, tol=1e-11)
        if it >= first:
            A, l, u = fq.get_constraints_coeffs(us, it)
            c = kkt_certificate(A, l, u, P, q, info, ro.cs.cuts, n_c=n_c, n_u=n_u, S=S, M=M, R=R, kappa=kappa,
                                alphaM=ro.cs.alphaM, saa=(method == "saa"), u_max=None)
            scale = max(1.0, c["multiplier_scale"])
            assert c["primal"] < 1e-9 and c["dual_sign"] < 1e-9 * scale, (it, c["primal"], c["dual_sign"])
            assert c["stationarity"] < 1e-9 * scale and c["complementarity"] < 1e-9 * scale, \
                (it, c["stationarity"], c["complementarity"])
            checked += 1
        us = nxt
    assert checked >= 4


def test_driving_ego_final_rows_closed_form_equals_the_oracle():
    """driving.Model.ego_final_rows (the sample-independent final rows of the table-free reduced solve, folded on the
    host in fp64) against the fp64 oracle's forward sensitivities (driving.py:283-288)."""
    import types
    from oracle import driving as ocar
    from riskaversetrajopt_amd import driving as drv
    for S in (2, 20, 40):
        samples = ocar.sample_uncertain_parameters(np.random.RandomState(0), 3, 'saa', S)
        o = ocar.Model(*samples, method='saa', alpha=0.2)
        t = np.arange(S)[:, None]
        uk = np.hstack([0.4 * np.cos(0.3 * t) + 0.1, 0.03 * np.sin(0.5 * t) + 0.004]) * (20.0 / S)
        fdu, flo, fup, _, _ = o.get_all_constraints_coeffs(uk)
        fake = types.SimpleNamespace(S=S, dt=ocar.T / S, _ego_init=ocar.state_init[:4].astype(np.float64))
        E, rhs = drv.Model.ego_final_rows(fake, uk)
        np.testing.assert_allclose(E, fdu[0], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(rhs, flo[0], rtol=1e-12, atol=1e-12)
        assert np.array_equal(fdu[0], fdu[-1])        # the ego carries no noise: the same rows for every sample


@pytest.mark.parametrize("system", ["drone", "driving"])
def test_streaming_fp64_leg_equals_the_dense_fp64_leg(system):
    """tests/_host_cuts.py: StreamingCutSolver (C oracle, one sample at a time: the fp64 leg at M = 1e5 in
    tests/test_gpu_scp.py) == HostCutSolver (NumPy, dense rows), subproblem by subproblem from the same iterate."""
    from tests._host_cuts import DroneStreamingOracle, DrivingStreamingOracle
    if system == "drone":
        o = _drone(60, 20, 0.1, "saa", seed=4)
        dense, stream, iters = DroneReducedOracle(o, delta=False), DroneStreamingOracle(o, nthreads=2), 6
    else:
        o = _car(48, 20, 0.1, "saa", seed=4)
        dense, stream, iters = DrivingReducedOracle(o, delta=False), DrivingStreamingOracle(o, nthreads=2), 5
    us = dense.initial_guess_us_mat()
    for k in range(iters):
        ud, td, idn = dense.solve_reduced(us, k)
        ust, tst, ist = stream.solve_reduced(us, k)
        assert abs(idn["cuts"] - ist["cuts"]) <= 2          # (the stopping test is a comparison against 1e-10)
        np.testing.assert_allclose(ust, ud, rtol=0, atol=1e-9)
        assert abs(tst - td) < 1e-9
        us = ud
