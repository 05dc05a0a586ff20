"""CPU: the exact dense master-QP solver (least-distance programming through NNLS)."""
import numpy as np
import pytest
from scipy.optimize import minimize

from riskaversetrajopt_amd import dense_qp


@pytest.mark.parametrize("seed", range(4))
def test_matches_slsqp_and_satisfies_kkt(seed):
    rng = np.random.RandomState(seed)
    n, me, mi = 20, 3, 30
    L = rng.randn(n, n)
    P = L @ L.T + 0.5 * np.eye(n)
    P[-1, -1] += 1e4                                   # badly scaled direction like the slack penalty
    q = rng.randn(n)
    q[-1] = 1e4
    x_feas = rng.randn(n)
    A_eq = rng.randn(me, n)
    b_eq = A_eq @ x_feas
    A_in = np.vstack([rng.randn(mi, n), np.eye(n), -np.eye(n)])
    b_in = np.concatenate([A_in[:mi] @ x_feas + rng.rand(mi), x_feas + 3, -(x_feas - 3)])
    x, lam = dense_qp.solve(P, q, A_eq, b_eq, A_in, b_in)
    assert np.max(np.abs(A_eq @ x - b_eq)) < 1e-9 and np.all(A_in @ x <= b_in + 1e-9) and np.all(lam >= 0)
    # stationarity on the equality-constrained subspace + complementarity
    g = P @ x + q + A_in.T @ lam
    nu = np.linalg.lstsq(A_eq.T, -g, rcond=None)[0]
    assert np.max(np.abs(g + A_eq.T @ nu)) < 1e-6 * max(1.0, np.abs(g).max())
    assert np.max(np.abs(lam * (A_in @ x - b_in))) < 1e-6
    cons = [{'type': 'eq', 'fun': lambda z: A_eq @ z - b_eq}, {'type': 'ineq', 'fun': lambda z: b_in - A_in @ z}]
    ref = minimize(lambda z: 0.5 * z @ P @ z + q @ z, x_feas, jac=lambda z: P @ z + q, constraints=cons,
                   method='SLSQP', options={'ftol': 1e-14, 'maxiter': 1000})
    assert 0.5 * x @ P @ x + q @ x <= ref.fun + 1e-6 * max(1.0, abs(ref.fun))
    np.testing.assert_allclose(x, ref.x, atol=2e-5)


def test_infeasible_master_is_reported():
    with pytest.raises(dense_qp.InfeasibleError):
        dense_qp.solve(np.eye(2), np.zeros(2), None, None, np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([-1.0, -1.0]))


def test_incremental_master_equals_one_shot():
    rng = np.random.RandomState(9)
    n = 15
    L = rng.randn(n, n)
    P = L @ L.T + np.eye(n)
    q = rng.randn(n)
    A_eq = rng.randn(2, n)
    xf = rng.randn(n)
    b_eq = A_eq @ xf
    A1, A2 = rng.randn(10, n), rng.randn(5, n)
    b1, b2 = A1 @ xf + rng.rand(10), A2 @ xf + rng.rand(5)
    m = dense_qp.Master(P, q, A_eq, b_eq)
    m.add_rows(A1, b1)
    x1, _ = m.solve()
    np.testing.assert_allclose(x1, dense_qp.solve(P, q, A_eq, b_eq, A1, b1)[0], atol=1e-9)
    m.add_rows(A2, b2)
    x2, lam = m.solve()
    np.testing.assert_allclose(x2, dense_qp.solve(P, q, A_eq, b_eq, np.vstack([A1, A2]), np.concatenate([b1, b2]))[0],
                               atol=1e-9)
    assert lam.shape == (15,) and np.all(lam >= 0)


def test_warm_started_nnls_equals_scipy_on_a_growing_problem():
    """The cutting-plane usage: columns are appended one at a time and every solve starts from the previous
    passive set; each solution must equal scipy's cold Lawson-Hanson solution."""
    from scipy.optimize import nnls
    rng = np.random.RandomState(4)
    for m, n_final in ((30, 60), (146, 120), (8, 40)):
        A = rng.randn(m, n_final)
        b = rng.randn(m)
        passive = np.zeros(0, dtype=bool)
        for n in range(1, n_final + 1, 3):
            y, passive, ok = dense_qp.nnls_warm(A[:, :n], b, passive)
            assert ok
            y_ref, _ = nnls(A[:, :n], b, maxiter=50 * n)
            # the fitted vector A y (projection of b on the cone) is unique; y itself only when the passive
            # columns are independent (n <= m here)
            np.testing.assert_allclose(A[:, :n] @ y, A[:, :n] @ y_ref, rtol=0, atol=1e-8 * max(1.0, np.abs(b).max()))
            if n <= m:
                np.testing.assert_allclose(y, y_ref, rtol=1e-7, atol=1e-9)
            assert np.all(y >= 0.0) and np.array_equal(passive, y > 0.0)
    # degenerate input: duplicate columns and a zero column
    A = rng.randn(12, 5)
    A = np.hstack([A, A[:, :2], np.zeros((12, 1))])
    b = rng.randn(12)
    y, _, ok = dense_qp.nnls_warm(A, b, np.ones(8, dtype=bool))
    y_ref, rn = nnls(A, b)
    assert ok and abs(np.linalg.norm(A @ y - b) - rn) < 1e-10


def test_master_warm_and_cold_paths_agree():
    rng = np.random.RandomState(12)
    n = 40
    L = rng.randn(n, n)
    Pm = L @ L.T + np.eye(n)
    q = rng.randn(n)
    A_eq = rng.randn(3, n)
    xf = rng.randn(n)
    b_eq = A_eq @ xf
    warm, cold = dense_qp.Master(Pm, q, A_eq, b_eq), dense_qp.Master(Pm, q, A_eq, b_eq)
    cold.warm = False
    for it in range(40):
        a = rng.randn(1, n)
        bb = a @ xf + rng.rand(1) * 0.5
        warm.add_rows(a, bb)
        cold.add_rows(a, bb)
        xw, lw = warm.solve()
        xc, lc = cold.solve()
        np.testing.assert_allclose(xw, xc, rtol=0, atol=1e-9)
        np.testing.assert_allclose(lw, lc, rtol=1e-6, atol=1e-9)


def test_master_diagonal_hessian_fast_path_equals_general_solver():
    """The SCP master has a diagonal Hessian: the whitened null-space path (no Cholesky) must give what the general
    dense solver gives."""
    rng = np.random.RandomState(21)
    n = 31
    d = rng.rand(n) * 5.0 + 0.1
    d[-1] = 1e4                                   # the slack penalty
    Pm = np.diag(d)
    q = np.zeros(n)
    q[-1] = 1e4
    A_eq = rng.randn(6, n)
    A_eq[:, -1] = 0.0
    xf = rng.randn(n)
    b_eq = A_eq @ xf
    m = dense_qp.Master(Pm, q, A_eq, b_eq)
    assert m.L is None                            # fast path taken
    rows, rhs = [], []
    for it in range(25):
        a = rng.randn(1, n)
        bb = a @ xf + rng.rand(1) * 0.3
        rows.append(a)
        rhs.append(bb)
        m.add_rows(a, bb)
        x, lam = m.solve()
        x_ref, lam_ref = dense_qp.solve(Pm + 1e-300 * np.ones((n, n)) * 0, q, A_eq, b_eq, np.vstack(rows), np.concatenate(rhs))
        np.testing.assert_allclose(x, x_ref, rtol=0, atol=1e-8)
        np.testing.assert_allclose(A_eq @ x, b_eq, atol=1e-9)
        assert np.all(np.vstack(rows) @ x <= np.concatenate(rhs) + 1e-9)


def test_native_nnls_matches_scipy_and_the_numpy_version():
    """rato_nnls_warm (csrc/nnls.hip: Lawson-Hanson with an incrementally updated QR, host code inside librato_saa.so)
    against scipy.optimize.nnls and the NumPy version it replaces in the master QP, cold and warm-started, incl. nearly
    dependent columns (consecutive cuts of a converging cutting-plane loop are nearly parallel)."""
    from scipy.optimize import nnls
    from riskaversetrajopt_amd import dense_qp
    assert dense_qp._native() is not None
    rng = np.random.RandomState(0)
    for trial in range(120):
        m, n = rng.randint(3, 160), rng.randint(1, 90)
        A = rng.randn(m, n)
        if trial % 3 == 0 and n > 1:
            A[:, n // 2:] = A[:, :n - n // 2] + 1e-7 * rng.randn(m, n - n // 2)
        b = rng.randn(m)
        y0, _ = nnls(A, b, maxiter=50 * n)
        P0 = (rng.rand(n) < 0.3) if trial % 2 else None
        y, P, ok = dense_qp.nnls_warm(A, b, P0)
        yp, Pp, okp = dense_qp.nnls_warm_py(A, b, P0)
        assert ok and np.all(y >= 0.0) and np.array_equal(P, y > 0.0)
        r0, r, rp = np.linalg.norm(A @ y0 - b), np.linalg.norm(A @ y - b), np.linalg.norm(A @ yp - b)
        assert r <= r0 * (1 + 1e-9) + 1e-12 and r <= rp * (1 + 1e-9) + 1e-12
        if trial % 3 and m >= 2 * n:                            # well conditioned and overdetermined: the minimiser is unique
            np.testing.assert_allclose(y, y0, rtol=1e-8, atol=1e-10)
    # KKT of the solution: dual w = A'(b - A y) <= tol on the zero set, = 0 on the passive set
    A, b = rng.randn(146, 70), rng.randn(146)
    y, P, ok = dense_qp.nnls_warm(A, b, None)
    w = A.T @ (b - A @ y)
    assert ok and np.all(w[~P] <= 1e-9) and np.all(np.abs(w[P]) <= 1e-9)


def test_native_master_equals_the_numpy_master():
    """rato_master_* (csrc/master.hip: Householder reflectors instead of a dense null-space basis, warm-started native
    NNLS) against MasterPy on an SCP-shaped problem: diagonal Hessian with a large slack penalty, 6 equalities, rows
    arriving one at a time -- same minimiser and multipliers after every row; infeasible rows are reported."""
    rng = np.random.RandomState(5)
    n = 151
    d = np.concatenate([np.tile([0.4, 0.4, 0.4], 50) * (1 + 0.1 * rng.rand(150)), [1e4]])
    q = np.zeros(n)
    q[-1] = 1e4
    A_eq = np.hstack([rng.randn(6, n - 1), np.zeros((6, 1))])
    xf = rng.randn(n)
    b_eq = A_eq @ xf
    nat, ref = dense_qp.Master(np.diag(d), q, A_eq, b_eq), dense_qp.MasterPy(np.diag(d), q, A_eq, b_eq)
    assert isinstance(nat, dense_qp.MasterNative)
    x, lam = nat.solve()                                   # no rows yet: the equality-constrained minimiser
    xr, _ = ref.solve()
    np.testing.assert_allclose(x, xr, rtol=0, atol=1e-10)
    assert np.abs(A_eq @ x - b_eq).max() < 1e-10
    for it in range(60):
        a = rng.randn(1, n)
        a[0, -1] = -3.0 if it % 2 else 0.0                 # cuts carry -c_s on the slack; bounds do not
        bb = a @ xf + rng.rand(1) * 0.5
        nat.add_rows(a, bb)
        ref.add_rows(a, bb)
        x, lam = nat.solve()
        xr, lamr = ref.solve()
        np.testing.assert_allclose(x, xr, rtol=0, atol=1e-8 * max(1.0, np.abs(xr).max()))
        np.testing.assert_allclose(lam, lamr, rtol=1e-6, atol=1e-8 * max(1.0, lamr.max()))
        assert lam.min() >= 0.0 and np.abs(A_eq @ x - b_eq).max() < 1e-9
    nat.add_rows(np.vstack([np.eye(n)[0], -np.eye(n)[0]]), np.array([-1.0, -1.0]))      # x_0 <= -1 and x_0 >= 1
    with pytest.raises(dense_qp.InfeasibleError):
        nat.solve()
    # rank-deficient equalities: the factory falls back to the NumPy master
    m2 = dense_qp.Master(np.diag(d), q, np.vstack([A_eq, A_eq[0]]), np.concatenate([b_eq, b_eq[:1]]))
    assert isinstance(m2, dense_qp.MasterPy)


def test_missing_library_is_loud(monkeypatch):
    """dense_qp drops to its NumPy master ONLY when the library cannot be loaded (OSError / RatoError), warns once, and
    raises under require_native() -- which scp.run_*_reduced switches on for device Models."""
    from riskaversetrajopt_amd import _lib, dense_qp
    def missing():
        raise _lib.RatoError("HIP extension is missing")
    monkeypatch.setattr(_lib, "load", missing)
    monkeypatch.setattr(dense_qp, "_NATIVE", {})
    monkeypatch.setattr(dense_qp, "_STRICT", [False])
    with pytest.warns(RuntimeWarning, match="not loadable"):
        assert dense_qp._native_master() is None
    assert dense_qp._native() is None                       # cached: no second warning, no second load
    dense_qp.require_native()
    with pytest.raises(_lib.RatoError):
        dense_qp._native_master()
    monkeypatch.setattr(dense_qp, "_NATIVE", {})
    with pytest.raises(_lib.RatoError):
        dense_qp._native()
    # anything else the loader throws is a bug and is not swallowed
    def broken():
        raise AttributeError("rato_master_create")
    monkeypatch.setattr(_lib, "load", broken)
    monkeypatch.setattr(dense_qp, "_NATIVE", {})
    monkeypatch.setattr(dense_qp, "_STRICT", [False])
    with pytest.raises(AttributeError):
        dense_qp._native_master()


def test_master_with_a_kept_factor_on_a_recorded_cutting_plane_sequence(monkeypatch):
    """The native master keeps the thin QR factor of the passive set between two solves (the columns of a growing NNLS
    problem do not change while its scale sigma does not).  Recorded sequence (tests/golden/master_sequence_*.npz: the rows
    a drone subproblem with M = 40, S = 20 added, in order, and the row counts at which it solved): near its end the cuts
    are nearly parallel and a sigma kept from an earlier, larger |v| made the NNLS cycle -- every solve must converge, and
    agree with a master that rescales (and refactors) at every solve."""
    import ctypes as C
    import os
    from riskaversetrajopt_amd import _lib
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "master_sequence_near_parallel_cuts.npz"))
    lib = _lib.load()

    def run():
        h = C.c_void_p()
        q, pd, A, b = (np.ascontiguousarray(f[k]) for k in ("q", "p_diag", "A_eq", "b_eq"))
        assert lib.rato_master_create(C.byref(h), q.shape[0], pd.ctypes.data, q.ctypes.data, A.shape[0], A.ctypes.data, b.ctypes.data) == 0
        rows, rhs = np.ascontiguousarray(f["rows"]), np.ascontiguousarray(f["rhs"])
        z, lam, done, out = np.zeros(q.shape[0]), np.zeros(rows.shape[0]), 0, []
        for n in f["solve_at"]:
            if n > done:
                assert lib.rato_master_add_rows(h, int(n - done), rows[done:n].ctypes.data, rhs[done:n].ctypes.data) == 0
                done = int(n)
            assert lib.rato_master_solve(h, z.ctypes.data, lam.ctypes.data) == 1, f"solve with {n} rows did not converge"
            assert np.all(rows[:done] @ z <= rhs[:done] + 1e-7 * (1.0 + np.abs(rhs[:done])))
            out.append(z.copy())
        lib.rato_master_destroy(h)
        return np.array(out)
    kept = run()
    # the reference behaviour: a fresh library state is not needed -- the stickiness is read once per process, so the
    # comparison leg is the NumPy master (rescales at every solve by construction)
    m = dense_qp.MasterPy(np.diag(f["p_diag"]), f["q"], f["A_eq"], f["b_eq"])
    done = 0
    for k, n in enumerate(f["solve_at"]):
        if n > done:
            m.add_rows(f["rows"][done:n], f["rhs"][done:n])
            done = int(n)
        x, _ = m.solve()
        np.testing.assert_allclose(kept[k], x, rtol=0, atol=2e-7 * max(1.0, np.abs(x).max()))
