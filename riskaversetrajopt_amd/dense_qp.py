"""Exact dense solver for the small strictly convex master QP of the cutting-plane loop.

    min 1/2 x' P x + q' x   s.t.  A_eq x = b_eq,   A_in x <= b_in          (P symmetric positive definite)

Method (Lawson & Hanson, "Solving Least Squares Problems", ch. 23): eliminate the equalities with a
null-space basis, whiten with the Cholesky factor of the reduced Hessian so that the objective becomes
1/2 |v|^2, and solve the resulting least-distance problem  min |v| s.t. G v >= h  through one
non-negative least squares (scipy.optimize.nnls, an active-set method: exact up to rounding, no
tolerances to tune, ~ms at the sizes used here: <= 3S+1 variables, a few hundred rows).
"""
import numpy as np
import scipy.linalg as sla
from scipy.optimize import nnls


class InfeasibleError(RuntimeError):
    pass


def solve(P, q, A_eq, b_eq, A_in, b_in):
    """-> (x, lam) with lam >= 0 the multipliers of the inequality rows."""
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    n = q.shape[0]
    A_in = np.asarray(A_in, dtype=np.float64).reshape(-1, n)
    b_in = np.asarray(b_in, dtype=np.float64)
    if A_eq is not None and len(A_eq):
        A_eq = np.asarray(A_eq, dtype=np.float64)
        x0 = np.linalg.lstsq(A_eq, np.asarray(b_eq, dtype=np.float64), rcond=None)[0]
        N = sla.null_space(A_eq)
    else:
        x0, N = np.zeros(n), np.eye(n)
    H = N.T @ P @ N
    c = N.T @ (P @ x0 + q)
    L = np.linalg.cholesky(H)                                   # H = L L'
    Linv_c = sla.solve_triangular(L, c, lower=True)
    # w = L^-T (v - L^-1 c);  rows: A_in (x0 + N w) <= b_in  ->  E v <= f
    E = sla.solve_triangular(L, (A_in @ N).T, lower=True).T     # A_in N L^-T
    f = b_in - A_in @ x0 + E @ Linv_c
    if E.shape[0] == 0:
        v = np.zeros(N.shape[1])
        lam = np.zeros(0)
    else:
        # scale rows (helps nnls): (E_i / s_i) v <= f_i / s_i
        sc = np.maximum(np.linalg.norm(E, axis=1), 1e-300)
        G, h = -(E / sc[:, None]), -(f / sc)
        A_n = np.vstack([G.T, h[None, :]])
        b_n = np.zeros(A_n.shape[0])
        b_n[-1] = 1.0
        y, _ = nnls(A_n, b_n, maxiter=20 * A_n.shape[1])
        r = A_n @ y - b_n
        if abs(r[-1]) < 1e-14:
            raise InfeasibleError("master QP infeasible")
        v = -r[:-1] / r[-1]
        lam = (y / (-r[-1])) / sc
    w = sla.solve_triangular(L.T, v - Linv_c, lower=False)
    return x0 + N @ w, lam


_NATIVE = {}
_STRICT = [False]


def require_native(on=True):
    """Product paths (scp.run_*_reduced on a device Model, bench.py) call this: from then on a library that cannot be
    loaded RAISES instead of dropping to the NumPy implementations below (a ~10x slower SCP would otherwise be timed
    without a word)."""
    _STRICT[0] = bool(on)


def _load_native():
    """librato_saa.so (host-side NNLS and master QP: csrc/nnls.hip, csrc/master.hip), or None when the library is not
    there -- ONLY then (OSError / RatoError from the loader; anything else is a bug and propagates), with one warning;
    under ``require_native()`` the loader's error is raised instead."""
    if "lib" not in _NATIVE:
        from . import _lib
        try:
            _NATIVE["lib"] = _lib.load()
        except (OSError, _lib.RatoError) as e:
            if _STRICT[0]:
                raise
            import warnings
            warnings.warn(f"librato_saa.so is not loadable ({e}): the master QP of the cutting-plane loop runs on its "
                          "NumPy implementation (~10x slower); product paths call dense_qp.require_native() and raise",
                          RuntimeWarning, stacklevel=3)
            _NATIVE["lib"] = None
    if _NATIVE["lib"] is None and _STRICT[0]:
        from . import _lib
        raise _lib.RatoError("librato_saa.so is required here (dense_qp.require_native) and could not be loaded")
    return _NATIVE["lib"]


def _native():
    """the library's rato_nnls_warm, or None (see ``_load_native``)"""
    lib = _load_native()
    return None if lib is None else lib.rato_nnls_warm


def nnls_warm(A, b, passive0=None, maxiter=None):
    """Lawson-Hanson NNLS, warm-started: the native implementation when the library is there, else ``nnls_warm_py``."""
    fn = _native()
    if fn is None:
        return nnls_warm_py(A, b, passive0, maxiter)
    m, n = A.shape
    Af = np.asfortranarray(A, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    P = np.zeros(n, dtype=np.uint8)
    if passive0 is not None and len(passive0):
        k = min(n, len(passive0))
        P[:k] = np.asarray(passive0[:k], dtype=bool)
    y = np.zeros(n)
    ok = fn(Af.ctypes.data, m, n, bb.ctypes.data, P.ctypes.data, y.ctypes.data, 0 if maxiter is None else int(maxiter))
    return y, P.astype(bool), ok == 1


def nnls_warm_py(A, b, passive0=None, maxiter=None):
    """Lawson-Hanson NNLS  min |A y - b|, y >= 0  started from a guess of the passive (positive) set.

    A cutting-plane loop solves a sequence of problems that differ by one column; the active-set method started
    from the previous passive set needs a handful of least-squares solves instead of one per positive variable.
    Termination is the KKT test of the original algorithm (dual w = A'(b - A y) <= tol on the zero set, y > 0 on
    the passive set; tol = 10 max(m,n) eps |A|_1 as in scipy).  -> (y, passive mask, converged)."""
    m, n = A.shape
    if maxiter is None:
        maxiter = 3 * n + 10
    tol = 10.0 * max(m, n) * np.finfo(np.float64).eps * max(np.abs(A).sum(axis=0).max(initial=0.0), 1e-300)
    y = np.zeros(n)
    P = np.zeros(n, dtype=bool)
    if passive0 is not None and len(passive0):
        k = min(n, len(passive0))
        P[:k] = np.asarray(passive0[:k], dtype=bool)
        for _ in range(n + 1):                      # warm start: keep the part of the guess that is positive
            if not P.any():
                break
            s = np.linalg.lstsq(A[:, P], b, rcond=None)[0]
            if np.all(s > 0.0):
                y[P] = s
                break
            idx = np.flatnonzero(P)
            P[idx[s <= 0.0]] = False
    for _ in range(maxiter):
        w = A.T @ (b - A @ y)
        w[P] = -np.inf
        j = int(np.argmax(w))
        if not (w[j] > tol):
            return y, P, True
        P[j] = True
        while True:
            idx = np.flatnonzero(P)
            s = np.linalg.lstsq(A[:, idx], b, rcond=None)[0]
            if np.all(s > 0.0):
                y[:] = 0.0
                y[idx] = s
                break
            neg = s <= 0.0
            yi = y[idx]
            denom = yi[neg] - s[neg]
            alpha = np.min(np.where(denom > 0.0, yi[neg] / np.where(denom > 0.0, denom, 1.0), 0.0))
            yi = yi + alpha * (s - yi)
            drop = neg & (yi <= 1e-300 + 0.0 * yi)   # the variables that hit zero leave the passive set
            if not drop.any():
                drop = neg                           # rounding: force progress
            yi[drop] = 0.0
            y[:] = 0.0
            y[idx] = yi
            P[idx[drop]] = False
            if not P.any():
                break
    return y, P, False


def _null_space(A):
    """Orthonormal basis of the null space of a wide full-row-rank A (m x n, m << n) from one complete QR of A^T (the
    SCP master: 6 x 151, tens of microseconds; scipy's SVD-based null_space took 0.23 ms of every SCP iteration);
    rank-deficient or ill-conditioned rows fall back to the SVD."""
    m, n = A.shape
    if m < n:
        Q, R = np.linalg.qr(A.T, mode='complete')
        d = np.abs(np.diagonal(R))
        if d.size and d.min() > 1e-10 * max(d.max(), 1e-300):
            return Q[:, m:]
    return sla.null_space(A)


def _native_master():
    """the library (rato_master_* entry points, csrc/master.hip), or None (see ``_load_native``)"""
    return _load_native()


def Master(P, q, A_eq, b_eq, p_diag=None):
    """The same solve with the equality elimination and the whitening done ONCE, and inequality rows appended
    incrementally (a cutting-plane loop adds one row per iteration).  Diagonal Hessian + full-row-rank equalities (the
    SCP master) -> the native implementation (``MasterNative``: csrc/master.hip, same algorithm, no dense null-space
    basis); anything else, or no library -> ``MasterPy``.  ``p_diag``: P is known to be diag(p_diag)."""
    P = np.asarray(P, dtype=np.float64)
    lib = _native_master()
    # ``p_diag``: the caller knows P = diag(p_diag) (the check costs 90 us on the SCP master: 9 % of a converged iteration)
    if lib is not None and A_eq is not None and len(A_eq) and (
            p_diag is not None or np.count_nonzero(P - np.diag(np.diagonal(P))) == 0):
        try:
            return MasterNative(lib, np.diagonal(P) if p_diag is None else p_diag, q, A_eq, b_eq)
        except ValueError:                         # rank-deficient equalities, non-positive diagonal
            pass
    return MasterPy(P, q, A_eq, b_eq)


class MasterNative:
    """ctypes face of rato_master_* (include/rato_saa.h)."""
    L = None                                       # (MasterPy's marker of the diagonal fast path)

    def __init__(self, lib, p_diag, q, A_eq, b_eq):
        import ctypes as C
        self._lib, self._C = lib, C
        p_diag = np.ascontiguousarray(p_diag, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        A_eq = np.ascontiguousarray(A_eq, dtype=np.float64)
        b_eq = np.ascontiguousarray(b_eq, dtype=np.float64)
        self.n = q.shape[0]
        h = C.c_void_p()
        rc = lib.rato_master_create(C.byref(h), self.n, p_diag.ctypes.data, q.ctypes.data, A_eq.shape[0],
                                    A_eq.ctypes.data, b_eq.ctypes.data)
        if rc != 0:
            raise ValueError(f"rato_master_create: status {rc}")
        self._h = h
        self._z = np.zeros(self.n)
        self.rows = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rato_master_destroy(h)

    def add_rows(self, A_in, b_in):
        A_in = np.ascontiguousarray(np.atleast_2d(np.asarray(A_in, dtype=np.float64)))
        b_in = np.ascontiguousarray(np.atleast_1d(np.asarray(b_in, dtype=np.float64)))
        if A_in.shape != (b_in.shape[0], self.n):
            raise ValueError(f"rows must be (k, {self.n}) with k right-hand sides, got {A_in.shape} / {b_in.shape}")
        rc = self._lib.rato_master_add_rows(self._h, A_in.shape[0], A_in.ctypes.data, b_in.ctypes.data)
        if rc != 0:
            raise ValueError(f"rato_master_add_rows: status {rc}")
        self.rows += A_in.shape[0]

    def solve(self):
        lam = np.zeros(self.rows)
        rc = self._lib.rato_master_solve(self._h, self._z.ctypes.data, lam.ctypes.data)
        if rc == -4:
            raise InfeasibleError("master QP infeasible")
        if rc != 1:
            raise RuntimeError(f"rato_master_solve: status {rc} (the NNLS of the master did not converge)")
        return self._z.copy(), lam


class MasterPy:
    """NumPy form of ``Master`` (general Hessian; also the reference the native one is tested against)."""

    def __init__(self, P, q, A_eq, b_eq):
        P = np.asarray(P, dtype=np.float64)
        q = np.asarray(q, dtype=np.float64)
        n = q.shape[0]
        if A_eq is not None and len(A_eq):
            A_eq = np.asarray(A_eq, dtype=np.float64)
            self.x0 = np.linalg.lstsq(A_eq, np.asarray(b_eq, dtype=np.float64), rcond=None)[0]
            self.N = None                      # one SVD per Master: computed in the branch that uses it
        else:
            self.x0, self.N = np.zeros(n), np.eye(n)
        if np.count_nonzero(P - np.diag(np.diagonal(P))) == 0 and A_eq is not None and len(A_eq):
            # diagonal Hessian (the SCP master: 2 dt R blocks and the slack penalty): whiten FIRST, x = D^-1/2 xt,
            # then the null-space basis of the scaled equalities is orthonormal in the whitened metric, the reduced
            # Hessian is the identity and no Cholesky factorization is needed
            d = 1.0 / np.sqrt(np.diagonal(P))
            Nt = _null_space(A_eq * d[None, :])                                  # orthonormal, (A_eq D^-1/2) Nt = 0
            self.N = d[:, None] * Nt
            self.L = None
            self.Linv_c = self.N.T @ (P @ self.x0 + q)                           # H = I
            self.NLinvT = self.N
        else:
            if self.N is None:
                self.N = _null_space(A_eq)
            H = self.N.T @ P @ self.N
            c = self.N.T @ (P @ self.x0 + q)
            self.L = np.linalg.cholesky(H)
            self.Linv_c = sla.solve_triangular(self.L, c, lower=True)
            self.NLinvT = sla.solve_triangular(self.L, self.N.T, lower=True).T      # N L^-T  (n x k)
        self.G = np.zeros((0, self.N.shape[1]))
        self.h = np.zeros(0)
        self.sc = np.zeros(0)
        self.passive = np.zeros(0, dtype=bool)      # passive set of the last solve (warm start of the next one)
        self.warm = True
        self.vnorm = 0.0                            # |v| of the last solve (scale of the next one)

    def add_rows(self, A_in, b_in):
        A_in = np.atleast_2d(np.asarray(A_in, dtype=np.float64))
        b_in = np.atleast_1d(np.asarray(b_in, dtype=np.float64))
        E = A_in @ self.NLinvT
        f = b_in - A_in @ self.x0 + E @ self.Linv_c
        sc = np.maximum(np.linalg.norm(E, axis=1), 1e-300)
        self.G = np.vstack([self.G, -(E / sc[:, None])])
        self.h = np.concatenate([self.h, -(f / sc)])
        self.sc = np.concatenate([self.sc, sc])

    def solve(self):
        if self.G.shape[0] == 0:
            v = np.zeros(self.N.shape[1])
            lam = np.zeros(0)
        else:
            # The NNLS residual's last component is 1/(1 + |v|^2): with |v| ~ 1e4 (an optimum far out, e.g. a large
            # slack) the KKT tolerance of the NNLS would accept constraint violations of tol (1 + |v|^2).  Solve for
            # v / sigma instead, sigma ~ |v| (previous solve, or the lower bound max_j h_j: rows are unit vectors).
            sigma = max(1.0, self.vnorm, float(self.h.max(initial=0.0)))
            A_n = np.vstack([self.G.T, (self.h / sigma)[None, :]])
            b_n = np.zeros(A_n.shape[0])
            b_n[-1] = 1.0
            ok = False
            if self.warm:
                y, P, ok = nnls_warm(A_n, b_n, self.passive)
                if ok:
                    self.passive = P
            if not ok:                               # cold fallback: scipy's Lawson-Hanson
                y, _ = nnls(A_n, b_n, maxiter=20 * A_n.shape[1])
                self.passive = y > 0.0
            r = A_n @ y - b_n
            if abs(r[-1]) < 1e-14:
                raise InfeasibleError("master QP infeasible")
            v = -sigma * r[:-1] / r[-1]
            lam = sigma * (y / (-r[-1])) / self.sc
            self.vnorm = float(np.linalg.norm(v))
        return self.x0 + self.NLinvT @ (v - self.Linv_c), lam
