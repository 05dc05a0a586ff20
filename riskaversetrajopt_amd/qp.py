"""Host QP solver with the call surface the reference uses from ``osqp``.

``osqp`` (an unpinned dependency of the reference, ``requirements.txt:3``) is not
installable in this image, so the convex subproblem of the SCP loop — which by
contract stays on the host CPU — is solved by this NumPy/SciPy restatement of
the published OSQP algorithm (Stellato et al., "OSQP: an operator splitting
solver for quadratic programs", 2020): Ruiz equilibration, ADMM with the
quasi-definite KKT system factorised once per rho, per-constraint rho
(equality rows x1e3), over-relaxation alpha = 1.6, adaptive rho, the standard
residual-based termination, primal-infeasibility detection and the polishing
step (active-set KKT solve with iterative refinement).

Only the calls the reference makes are provided (``drone_risk.py:433-457``,
``driving.py:430-444``): ``OSQP().setup(P, q, A, l, u, eps_abs=, eps_rel=,
linsys_solver=, warm_start=, verbose=, polish=)``, ``update(l=, u=)``,
``update(Ax=)``, ``solve()`` -> result with ``.x``, ``.y``, ``.info.status``.
When polishing succeeds the returned point is the exact optimum of the QP for the
identified active set, which is what makes SCP iterates comparable at 1e-5.
"""
import time

import numpy as np
import scipy.sparse as sp
import scipy.linalg as sla
import scipy.sparse.linalg as spla

OSQP_INFTY = 1e30
RHO_MIN, RHO_MAX = 1e-6, 1e6
RHO_EQ_OVER_RHO_INEQ = 1e3
RHO_TOL = 1e-4


class _Info:
    status = "unsolved"
    status_val = 0
    iter = 0
    obj_val = np.nan
    pri_res = np.nan
    dua_res = np.nan
    setup_time = 0.0
    solve_time = 0.0
    polish_time = 0.0
    run_time = 0.0
    status_polish = 0
    rho_updates = 0
    rho_estimate = np.nan


class _Result:
    def __init__(self):
        self.x = None
        self.y = None
        self.info = _Info()


def _inf_norm_cols(Msp):
    """max |entry| per column of a sparse matrix (0 for empty columns)."""
    if Msp.shape[0] == 0 or Msp.nnz == 0:
        return np.zeros(Msp.shape[1])
    return np.asarray(abs(Msp).max(axis=0).todense()).ravel()


class OSQP:
    def __init__(self):
        self._setup_done = False

    # ------------------------------------------------------------------ setup
    def setup(self, P=None, q=None, A=None, l=None, u=None, eps_abs=1e-3, eps_rel=1e-3, rho=0.1, sigma=1e-6,
              alpha=1.6, max_iter=4000, scaling=10, adaptive_rho=True, adaptive_rho_interval=50,
              adaptive_rho_tolerance=5.0, polish=False, polish_refine_iter=3, delta=1e-6, warm_start=True,
              verbose=False, linsys_solver=None, eps_prim_inf=1e-4, check_termination=25, polish_retry=True,
              linsys="auto", **_ignored):
        t0 = time.perf_counter()
        self.n, self.m = P.shape[0], A.shape[0]
        self.P = sp.csc_matrix(P, dtype=np.float64)
        self.P = sp.triu(self.P, format="csc") + sp.triu(self.P, k=1, format="csc").T     # symmetrise like osqp
        self.P = sp.csc_matrix(self.P)
        self.q = np.asarray(q, dtype=np.float64).copy()
        self.A = sp.csc_matrix(A, dtype=np.float64)
        self.A.sort_indices()
        self.l = np.maximum(np.nan_to_num(np.asarray(l, dtype=np.float64), nan=-OSQP_INFTY), -OSQP_INFTY)
        self.u = np.minimum(np.nan_to_num(np.asarray(u, dtype=np.float64), nan=OSQP_INFTY), OSQP_INFTY)
        if np.any(self.l > self.u):
            raise ValueError("lower bound must be lower than or equal to upper bound")
        self.opts = dict(eps_abs=eps_abs, eps_rel=eps_rel, rho=rho, sigma=sigma, alpha=alpha, max_iter=max_iter,
                         scaling=scaling, adaptive_rho=adaptive_rho, adaptive_rho_interval=adaptive_rho_interval,
                         adaptive_rho_tolerance=adaptive_rho_tolerance, polish=polish,
                         polish_refine_iter=polish_refine_iter, delta=delta, warm_start=warm_start,
                         verbose=verbose, eps_prim_inf=eps_prim_inf, check_termination=check_termination,
                         polish_retry=polish_retry, linsys=linsys)
        self.rho = float(rho)
        self._scale()
        self._make_rho_vec()
        self._factorize()
        self.x = np.zeros(self.n)
        self.z = np.zeros(self.m)
        self.y = np.zeros(self.m)
        self._setup_done = True
        self._setup_time = time.perf_counter() - t0
        return self

    def _scale(self):
        """Ruiz equilibration of [[P, A'], [A, 0]] + cost scaling (OSQP §5.1)."""
        n, m = self.n, self.m
        D, E, c = np.ones(n), np.ones(m), 1.0
        P, A, q = self.P.copy(), self.A.copy(), self.q.copy()
        for _ in range(int(self.opts["scaling"])):
            dcol = np.maximum(_inf_norm_cols(P), _inf_norm_cols(A))
            ecol = _inf_norm_cols(A.T.tocsc())
            dcol = np.where(dcol < 1e-4, 1.0, dcol)
            ecol = np.where(ecol < 1e-4, 1.0, ecol)
            d = 1.0 / np.sqrt(np.minimum(dcol, 1e4))
            e = 1.0 / np.sqrt(np.minimum(ecol, 1e4))
            Dm, Em = sp.diags(d), sp.diags(e)
            P = (Dm @ P @ Dm).tocsc()
            A = (Em @ A @ Dm).tocsc()
            q = d * q
            D, E = D * d, E * e
            pn = _inf_norm_cols(P)
            cost = max(float(np.mean(pn)) if n else 0.0, float(np.max(np.abs(q))) if n else 0.0)
            cost = 1.0 if cost < 1e-4 else min(cost, 1e4)
            g = 1.0 / cost
            P, q, c = P * g, q * g, c * g
        self.D, self.E, self.c = D, E, c
        self.Ps, self.As, self.qs = sp.csc_matrix(P), sp.csc_matrix(A), q
        self.ls, self.us = self.l * E, self.u * E
        self.ls[self.l <= -OSQP_INFTY] = -OSQP_INFTY
        self.us[self.u >= OSQP_INFTY] = OSQP_INFTY

    def _make_rho_vec(self):
        eq = np.abs(self.ls - self.us) < RHO_TOL
        free = (self.ls <= -OSQP_INFTY) & (self.us >= OSQP_INFTY)
        rv = np.full(self.m, self.rho)
        rv[eq] = min(RHO_EQ_OVER_RHO_INEQ * self.rho, RHO_MAX)
        rv[free] = RHO_MIN
        self.rho_vec = rv

    # The SAA QPs are tall: m ~ 3 S M rows against n = n_u S + M + 2 columns (M = 500, S = 20: 30,568 x 562).  The
    # quasi-definite KKT matrix [[P + sigma I, A'], [A, -R^-1]] that osqp's direct solver (and the first version of this
    # class) factorises has n + m rows; eliminating nu = R (A x - z) + y leaves the n x n SPD system
    #     (P + sigma I + A' R A) x = sigma x_k - q + A' (R z_k - y_k)
    # -- osqp's "indirect" form, solved here with a dense Cholesky factorisation (n is a few hundred to a few thousand).
    # Same iterates as the KKT form up to rounding (tests/test_qp.py runs both).
    REDUCED_MAX_N = 6000

    def _use_reduced(self):
        mode = self.opts.get("linsys", "auto")
        if mode == "kkt":
            return False
        if mode == "reduced":
            return True
        return self.m > 2 * self.n and self.n <= self.REDUCED_MAX_N

    def _factorize(self):
        n, m = self.n, self.m
        if self._use_reduced():
            R = sp.diags(self.rho_vec)
            H = (self.Ps + (self.As.T @ (R @ self.As))).toarray()
            H[np.diag_indices(n)] += self.opts["sigma"]
            self._chol = sla.cho_factor(H, lower=True, overwrite_a=True, check_finite=False)
            self._lu = None
            return
        K = sp.bmat([[self.Ps + self.opts["sigma"] * sp.eye(n), self.As.T],
                     [self.As, -sp.diags(1.0 / self.rho_vec)]], format="csc")
        self._lu = spla.splu(K)
        self._chol = None

    def _kkt_solve(self, xs, zs, ys):
        """-> (x_tilde, nu) of the ADMM linear system at (xs, zs, ys)."""
        sigma = self.opts["sigma"]
        if self._chol is not None:
            rhs = sigma * xs - self.qs + self.As.T @ (self.rho_vec * zs - ys)
            xt = sla.cho_solve(self._chol, rhs, check_finite=False)
            nu = self.rho_vec * (self.As @ xt - zs) + ys
            return xt, nu
        rhs = np.concatenate([sigma * xs - self.qs, zs - ys / self.rho_vec])
        sol = self._lu.solve(rhs)
        return sol[:self.n], sol[self.n:]

    # ----------------------------------------------------------------- update
    def update(self, q=None, l=None, u=None, Px=None, Px_idx=None, Ax=None, Ax_idx=None):
        if not self._setup_done:
            raise RuntimeError("setup() first")
        refactor = False
        if q is not None:
            self.q = np.asarray(q, dtype=np.float64).copy()
        if l is not None:
            self.l = np.maximum(np.nan_to_num(np.asarray(l, dtype=np.float64), nan=-OSQP_INFTY), -OSQP_INFTY)
        if u is not None:
            self.u = np.minimum(np.nan_to_num(np.asarray(u, dtype=np.float64), nan=OSQP_INFTY), OSQP_INFTY)
        if Ax is not None:
            Ax = np.asarray(Ax, dtype=np.float64)
            if Ax_idx is None:
                if Ax.shape[0] != self.A.nnz:
                    raise ValueError("update(Ax=...): new values do not match the sparsity pattern given to setup()")
                self.A.data[:] = Ax
            else:
                self.A.data[Ax_idx] = Ax
            refactor = True
        if Px is not None:
            raise NotImplementedError("update(Px=...) is not used by the reference")
        if refactor or q is not None:
            # like osqp's update_A / update_lin_cost: keep the equilibration (D, E, c) computed at setup,
            # re-apply it to the new values and refactorise; the iterates are kept for the warm start
            self.As = sp.csc_matrix(sp.diags(self.E) @ self.A @ sp.diags(self.D))
            self.qs = self.c * self.D * self.q
            self.ls, self.us = self.l * self.E, self.u * self.E
            self.ls[self.l <= -OSQP_INFTY] = -OSQP_INFTY
            self.us[self.u >= OSQP_INFTY] = OSQP_INFTY
            self._make_rho_vec()
            self._factorize()
        else:
            self.ls, self.us = self.l * self.E, self.u * self.E
            self.ls[self.l <= -OSQP_INFTY] = -OSQP_INFTY
            self.us[self.u >= OSQP_INFTY] = OSQP_INFTY
            old = self.rho_vec.copy()
            self._make_rho_vec()
            if not np.array_equal(old, self.rho_vec):
                self._factorize()
        return self

    def warm_start(self, x=None, y=None):
        if x is not None:
            self.x = np.asarray(x, dtype=np.float64).copy()
            self.z = self.A @ self.x
        if y is not None:
            self.y = np.asarray(y, dtype=np.float64).copy()

    # ------------------------------------------------------------------ solve
    def _residuals(self, xs, zs, ys):
        """Unscaled residuals and the norms entering the termination test."""
        Einv, Dinv = 1.0 / self.E, 1.0 / self.D
        Ax = self.As @ xs
        Px = self.Ps @ xs
        Aty = self.As.T @ ys
        r_prim = np.max(np.abs(Einv * (Ax - zs))) if self.m else 0.0
        r_dual = np.max(np.abs(Dinv * (Px + self.qs + Aty))) / self.c
        n_prim = max(np.max(np.abs(Einv * Ax)) if self.m else 0.0, np.max(np.abs(Einv * zs)) if self.m else 0.0)
        n_dual = max(np.max(np.abs(Dinv * Px)), np.max(np.abs(Dinv * Aty)) if self.m else 0.0,
                     np.max(np.abs(Dinv * self.qs))) / self.c
        return r_prim, r_dual, n_prim, n_dual

    def solve(self):
        """ADMM + polish.  With ``polish_retry`` (default when polish=True): if polishing fails at the
        requested tolerance, ADMM continues with a 10x tighter tolerance (up to 3 times) and polishing
        is retried, so that the returned point is the QP optimum whenever an active set can be
        identified — the reference's eps = 1e-3 iterate alone is not unique enough to compare SCP
        iterates across back-ends (its dual tolerance is relative to ||q||_inf = 1e4)."""
        o = self.opts
        res = self._solve_once(o["eps_abs"], o["eps_rel"])
        tries = 0
        while (o["polish"] and o.get("polish_retry", True) and res.info.status == "solved"
               and res.info.status_polish != 1 and tries < 3):
            tries += 1
            f = 10.0 ** (-tries)
            prev = res
            res = self._solve_once(o["eps_abs"] * f, o["eps_rel"] * f)
            res.info.iter += prev.info.iter
            res.info.run_time += prev.info.run_time
            if res.info.status != "solved":      # the tighter pass ran out of iterations: keep the
                prev.info.iter, prev.info.run_time = res.info.iter, res.info.run_time   # eps-accurate point
                self.x, self.y = prev.x, prev.y
                self.z = self.A @ prev.x
                return prev
        return res

    def _solve_once(self, eps_abs, eps_rel):
        o = self.opts
        t0 = time.perf_counter()
        res = _Result()
        n, m = self.n, self.m
        sigma, alpha = o["sigma"], o["alpha"]
        if o["warm_start"]:
            xs, zs, ys = self.x / self.D, self.z * self.E, self.y / self.E * self.c
        else:
            xs, zs, ys = np.zeros(n), np.zeros(m), np.zeros(m)
        status = "maximum iterations reached"
        it = 0
        for it in range(1, o["max_iter"] + 1):
            xt, nu = self._kkt_solve(xs, zs, ys)
            zt = zs + (nu - ys) / self.rho_vec
            x_new = alpha * xt + (1 - alpha) * xs
            z_relax = alpha * zt + (1 - alpha) * zs
            z_new = np.minimum(np.maximum(z_relax + ys / self.rho_vec, self.ls), self.us)
            dy = self.rho_vec * (z_relax - z_new)
            y_new = ys + dy
            dx = x_new - xs
            xs, zs, ys = x_new, z_new, y_new
            if it % o["check_termination"] == 0 or it == o["max_iter"]:   # like osqp: first check at iteration 25
                r_prim, r_dual, n_prim, n_dual = self._residuals(xs, zs, ys)
                eps_p = eps_abs + eps_rel * n_prim
                eps_d = eps_abs + eps_rel * n_dual
                if r_prim <= eps_p and r_dual <= eps_d:
                    status = "solved"
                    break
                # primal infeasibility certificate (OSQP §3.4) on the dual increment
                dyu = self.E * dy
                nrm = np.max(np.abs(dyu)) if m else 0.0
                if nrm > 1e-12:
                    dyn = dyu / nrm
                    if (np.max(np.abs(self.D * (self.As.T @ (dy / nrm)))) <= o["eps_prim_inf"] and
                            (self.u @ np.maximum(dyn, 0) * (1) + self.l @ np.minimum(dyn, 0)) <= -o["eps_prim_inf"]
                            and np.all(np.isfinite(self.u[dyn > 0])) and np.all(np.isfinite(self.l[dyn < 0]))):
                        status = "primal infeasible"
                        break
                if o["adaptive_rho"] and it % o["adaptive_rho_interval"] == 0:
                    num = r_prim / max(n_prim, 1e-10)
                    den = r_dual / max(n_dual, 1e-10)
                    new_rho = float(np.clip(self.rho * np.sqrt(num / max(den, 1e-10)), RHO_MIN, RHO_MAX))
                    if new_rho > self.rho * o["adaptive_rho_tolerance"] or new_rho < self.rho / o["adaptive_rho_tolerance"]:
                        self.rho = new_rho
                        self._make_rho_vec()
                        self._factorize()
                        res.info.rho_updates += 1
        r_prim, r_dual, _, _ = self._residuals(xs, zs, ys)
        x = self.D * xs
        y = self.E * ys / self.c
        z = zs / self.E
        self.x, self.y, self.z = x, y, z
        res.info.solve_time = time.perf_counter() - t0
        res.info.status = status
        res.info.status_val = 1 if status == "solved" else (-3 if status == "primal infeasible" else -2)
        res.info.iter = it
        res.info.pri_res, res.info.dua_res = r_prim, r_dual
        res.info.rho_estimate = self.rho
        if status == "solved" and o["polish"]:
            tp = time.perf_counter()
            ok, xp, yp = self._polish(x, y, z)
            res.info.status_polish = 1 if ok else -1
            if ok:
                x, y = xp, yp
                self.x, self.y, self.z = x, y, self.A @ x
            res.info.polish_time = time.perf_counter() - tp
        res.x, res.y = x, y
        if status == "primal infeasible":
            res.x = np.full(n, np.nan)
        res.info.obj_val = float(0.5 * x @ (self.P @ x) + self.q @ x)
        res.info.setup_time = self._setup_time
        res.info.run_time = res.info.solve_time + res.info.polish_time
        return res

    def _polish(self, x, y, z):
        """OSQP §4: solve the equality-constrained QP on the active set guessed from y."""
        o = self.opts
        delta = o["delta"]
        lo = (z - self.l < -y)            # y < 0 and close to the lower bound
        up = (self.u - z < y)             # y > 0 and close to the upper bound
        A_L, A_U = self.A[np.nonzero(lo)[0], :], self.A[np.nonzero(up)[0], :]
        nL, nU = A_L.shape[0], A_U.shape[0]
        n = self.n
        blocks = [[self.P, A_L.T, A_U.T], [A_L, None, None], [A_U, None, None]]
        K = sp.bmat([[b for b in row] for row in blocks], format="csc") if (nL + nU) else sp.csc_matrix(self.P)
        Kreg = K + sp.diags(np.concatenate([np.full(n, delta), np.full(nL + nU, -delta)]))
        rhs = np.concatenate([-self.q, self.l[lo], self.u[up]])
        try:
            lu = spla.splu(sp.csc_matrix(Kreg))
        except RuntimeError:
            return False, x, y
        sol = lu.solve(rhs)
        for _ in range(o["polish_refine_iter"]):           # iterative refinement on the unregularised system
            sol = sol + lu.solve(rhs - K @ sol)
        xp = sol[:n]
        yp = np.zeros(self.m)
        yp[lo] = sol[n:n + nL]
        yp[up] = sol[n + nL:]
        zp = self.A @ xp
        pri = max(np.max(np.maximum(self.l - zp, 0.0), initial=0.0), np.max(np.maximum(zp - self.u, 0.0), initial=0.0))
        dua = np.max(np.abs(self.P @ xp + self.q + self.A.T @ yp))
        pri0 = max(np.max(np.maximum(self.l - self.A @ x, 0.0), initial=0.0),
                   np.max(np.maximum(self.A @ x - self.u, 0.0), initial=0.0))
        dua0 = np.max(np.abs(self.P @ x + self.q + self.A.T @ y))
        ok = np.all(np.isfinite(xp)) and ((pri <= pri0 and dua <= dua0) or (pri <= pri0 + 1e-10 and dua < 1e-10) or
                                         (pri < 1e-10 and dua < 1e-10))
        return bool(ok), xp, yp
