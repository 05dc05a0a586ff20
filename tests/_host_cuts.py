"""Test helper: the reduced (u, slack) SCP subproblem with a HOST fp64 cut oracle.

``riskaversetrajopt_amd.cvar_cuts.CvarCutSolver`` owns the cutting-plane loop and the exact master QP; its two
oracle methods run on the device.  Here they are restated in NumPy fp64 on the fp64 oracle's dense linearization
(``oracle.drone`` / ``oracle.driving``), so that
  * the loop itself (elimination of y / t, 'baseline' mode, the relaxed first iterations, cut recycling) can be
    checked against the full QP on the CPU, and
  * the device path can be compared with an all-fp64 run of the SAME algorithm at batch sizes no host QP can take.
Checker only: nothing in the package imports this."""
import numpy as np

from riskaversetrajopt_amd import cvar_cuts


class HostCutSolver(cvar_cuts.CvarCutSolver):
    """evaluate / relinearize_kept_cuts on dense fp64 rows:  G (M, R*S, nU),  base (M, R*S)."""

    def __init__(self, **kw):
        super().__init__(None, None, **kw)
        self.cuts = {}                    # slot -> (weights (M,), arg-max rows (M,))

    def _weights(self, m):
        """tail weights of CVaR_alpha: 1 above the threshold, the remaining mass spread evenly over the ties with it"""
        aM = self.alphaM
        order = np.sort(m)[::-1]
        t = order[min(int(np.ceil(aM - 1e-12)) - 1, m.size - 1)]
        gt, eq = m > t, m == t
        lam = min(max((aM - gt.sum()) / max(eq.sum(), 1), 0.0), 1.0)
        # reported VaR: the reference's sort(Z)[M - floor(alpha M) - 1] (drone_main_plot.py:649-651) -- when alpha M is
        # an integer every t between that value and the next larger one minimises the Rockafellar-Uryasev function, and
        # the QP leaves t_risk anywhere in that interval; the device reports the same end of it (rato_risk_stats out[0])
        var = order[min(int(np.floor(aM + 1e-12)), m.size - 1)] if self.mode == 'saa' else t
        return gt * 1.0 + eq * lam, float(var)

    def evaluate(self, G, W, tile, base, u_vec, slot=None):
        sign, x0 = self._form()
        x = np.asarray(u_vec, dtype=np.float64) - x0
        rows = G @ x + sign * base                                   # (M, R*S)
        arg = rows.argmax(axis=1)
        m = rows[np.arange(rows.shape[0]), arg]
        w, t = self._weights(m)
        if slot is not None:
            self.cuts[slot] = (w, arg)
        idx = np.arange(rows.shape[0])
        g = (w[:, None] * G[idx, arg]).sum(axis=0) / self.alphaM
        phi = float(w @ m) / self.alphaM
        return phi, t, g

    def relinearize_kept_cuts(self, G, W, tile, base):
        K = len(self.keep)
        if K == 0:
            return np.zeros((0, self.nU)), np.zeros(0)
        sign, x0 = self._form()
        idx = np.arange(G.shape[0])
        rows, rhs = np.zeros((K, self.nU)), np.zeros(K)
        for k, slot in enumerate(self.keep):
            w, arg = self.cuts[slot]
            rows[k] = (w[:, None] * G[idx, arg]).sum(axis=0) / self.alphaM
            rhs[k] = self.rhs0 + rows[k] @ x0 - sign * float(w @ base[idx, arg]) / self.alphaM
        return rows, rhs


class _ReducedOracleModel:
    """``solve_reduced`` / ``initial_guess_us_mat`` on the fp64 oracle: what scp.run_*_reduced drives."""

    def __init__(self, om, n_u, R, u_max, Rcost, slack_penalty, rhs0, first_cvar_iter, delta=True):
        self.o, self.n_u, self.R, self.delta, self.first_cvar_iter = om, n_u, R, delta, first_cvar_iter
        self.cs = HostCutSolver(n_u=n_u, S=om.S, M=om.M, ld=om.M, R=R, alpha=om.alpha, dt=om.dt, Rcost=Rcost,
                                slack_penalty=slack_penalty, u_min=-u_max, u_max=u_max, mode=om.method, rhs0=rhs0)

    def initial_guess_us_mat(self):
        return self.o.initial_guess_us_mat()

    def solve_reduced(self, us, scp_iter, tol=1e-10):
        fdu, frhs, gdu, gup = self.linearization(us)
        M, nU = self.o.M, self.n_u * self.o.S
        G = gdu.reshape(M, -1, nU)
        gup = gup.reshape(M, -1)
        if self.delta:
            base, u_lin = -(gup - G @ np.asarray(us, dtype=np.float64).reshape(-1)), us     # g = -(g_up - G u_k)
        else:
            base, u_lin = gup, None
        info = self.cs.solve(G, None, 0, base, fdu, frhs, u_lin=u_lin, with_cvar=(scp_iter >= self.first_cvar_iter),
                             tol=tol)
        return info["us"], info["t_risk"], info


class DroneReducedOracle(_ReducedOracleModel):
    def __init__(self, om, delta=True):
        from oracle import drone as od
        super().__init__(om, 3, 3, od.u_max, od.R, 10000.0, -1e-3 / 0.01, 2, delta)

    def linearization(self, us):
        fdu, flo, _, gdu, gup = self.o.get_all_constraints_coeffs(us)
        return fdu.mean(0), flo.mean(0), gdu, gup


class DrivingReducedOracle(_ReducedOracleModel):
    def __init__(self, om, delta=True):
        from oracle import driving as ocar
        super().__init__(om, 2, 1, ocar.u_max, ocar.R, 1000.0, 0.0, 1, delta)

    def linearization(self, us):
        fdu, flo, _, gdu, gup = self.o.get_all_constraints_coeffs(us)
        return fdu[0], flo[0], gdu, gup


class StreamingCutSolver(HostCutSolver):
    """The same two oracle methods on the oracle's C streaming cut oracle (oracle/saa_oracle.c, fp64, OpenMP): one
    sample's dense linearization is formed at a time and nothing is stored per sample, so the all-fp64 leg runs at the
    batch sizes the benchmark is quoted on (M = 1e5: the dense rows would be 18 GB).  Reference form of the rows
    (G u - g_up, drone_risk.py:357-364): no linearization point enters the algebra."""

    def __init__(self, cut_oracle, **kw):
        super().__init__(**kw)
        self.co, self.us_k = cut_oracle, None

    def evaluate(self, G, W, tile, base, u_vec, slot=None):
        u = np.asarray(u_vec, dtype=np.float64)
        m, arg = self.co.rowmax(self.us_k, u)
        w, t = self._weights(m)
        if slot is not None:
            self.cuts[slot] = (w, arg)
        grad, _ = self.co.tail_rows(self.us_k, w, arg)
        return float(w @ m) / self.alphaM, t, grad[0] / self.alphaM

    def relinearize_kept_cuts(self, G, W, tile, base):
        K = len(self.keep)
        if K == 0:
            return np.zeros((0, self.nU)), np.zeros(0)
        w = np.stack([self.cuts[sl][0] for sl in self.keep])
        arg = np.stack([self.cuts[sl][1] for sl in self.keep])
        grad, gup = self.co.tail_rows(self.us_k, w, arg)
        return grad / self.alphaM, self.rhs0 + gup / self.alphaM


class _StreamingReducedModel:
    """``solve_reduced`` / ``initial_guess_us_mat`` with the streaming fp64 cut oracle (any M)."""

    def __init__(self, om, co, n_u, R, u_max, Rcost, slack_penalty, rhs0, first_cvar_iter):
        self.o, self.co, self.first_cvar_iter = om, co, first_cvar_iter
        self.cs = StreamingCutSolver(co, n_u=n_u, S=om.S, M=om.M, ld=om.M, R=R, alpha=om.alpha, dt=om.dt, Rcost=Rcost,
                                     slack_penalty=slack_penalty, u_min=-u_max, u_max=u_max, mode=om.method, rhs0=rhs0)

    def initial_guess_us_mat(self):
        return self.o.initial_guess_us_mat()

    def solve_reduced(self, us, scp_iter, tol=1e-10):
        fdu, frhs = self.final_rows(us)
        self.cs.us_k = np.asarray(us, dtype=np.float64).copy()
        info = self.cs.solve(None, None, 0, None, fdu, frhs, u_lin=None,
                             with_cvar=(scp_iter >= self.first_cvar_iter), tol=tol)
        return info["us"], info["t_risk"], info


class DroneStreamingOracle(_StreamingReducedModel):
    def __init__(self, om, nthreads=0):
        from oracle import c_oracle, drone as od
        co = c_oracle.DroneCutOracle(om.DWs, om.masses, om.obs_Qs, om.dt, nthreads)
        super().__init__(om, co, 3, 3, od.u_max, od.R, 10000.0, -1e-3 / 0.01, 2)

    def final_rows(self, us):
        return self.co.final_rows(us)


class DrivingStreamingOracle(_StreamingReducedModel):
    def __init__(self, om, nthreads=0):
        from oracle import c_oracle, driving as ocar
        co = c_oracle.CarCutOracle(om.states_init, om.omegas_speed, om.omegas_repulsive, om.DWs, nthreads)
        super().__init__(om, co, 2, 1, ocar.u_max, ocar.R, 1000.0, 0.0, 1)
        # the final rows only involve the ego car, which is sample-independent (driving.py:311-313 averages M equal
        # rows): one sample of the NumPy oracle gives them
        self._one = ocar.Model(om.states_init[:1], om.omegas_speed[:1], om.omegas_repulsive[:1], om.DWs[:1],
                               method=om.method, alpha=om.alpha)

    def final_rows(self, us):
        fdu, flo, _, _, _ = self._one.get_all_constraints_coeffs(us)
        return fdu[0], flo[0]


def kkt_certificate(A, l, u, P, q, info, cut_data, *, n_c, n_u, S, M, R, kappa, alphaM, saa, u_max):
    """Optimality certificate of a reduced solution against the reference's FULL QP (its own row and column layout:
    SURVEY appendix A; ``A, l, u`` from ``get_constraints_coeffs`` / ``assemble.saa_constraints``, ``P, q`` from
    ``get_objective_coeffs``).  The reduced solution is lifted to z = (u, y, slack, t) with y_i = max(-slack, m_i - t)
    and the master's multipliers are spread over the full QP's rows:
        cut k (multiplier lam_k, tail weights w_ki, rows r_ki):   kappa rho_{i, r_ki} += lam_k w_ki / (alpha M)
        CVaR row:  mu = sum_k lam_k / (M alpha);   -y_i - slack rows:  pi_i = mu - kappa sum_r rho_ir;   -slack row: sigma
        control bounds: the master's own bound multipliers;   final rows: least squares on the u-stationarity.
    -> dict of the KKT residuals (all 0 at the optimum): primal feasibility, stationarity P z + q + A'y, dual sign,
    complementarity.  ``cut_data[slot]`` = (weights (M,), arg-max rows (M,)) of every cut with a multiplier."""
    nU, R_s = n_u * S, R * S
    us = np.asarray(info["us"], dtype=np.float64).reshape(-1)
    s = float(info["slack"])
    mult = info["multipliers"]
    assert mult["uncertified_cuts"] == 0
    n_head = (1 + M) if saa else 0
    obs0 = n_c + n_head
    Aobs = A[obs0:obs0 + M * R_s]
    rows_val = (Aobs[:, :nU] @ us - u[obs0:obs0 + M * R_s]) / kappa                    # (G_i u - g_up_i)_r (+ pad / kappa)
    m_i = rows_val.reshape(M, R_s).max(axis=1)
    if saa:
        t = float(info["t_risk"])
        y = np.maximum(-s, m_i - t)
    else:
        t, y = 0.0, np.zeros(M)
    z = np.concatenate([us, y, [s, t]])
    ydual = np.zeros(A.shape[0])
    Lam = 0.0
    for slot, lam in mult["cuts"]:
        if lam <= 0.0:
            continue
        w, arg = cut_data[slot]
        ydual[obs0 + np.arange(M) * R_s + arg] += lam * w / (alphaM * kappa)
        Lam += lam
    if saa:
        mu = Lam / alphaM
        ydual[n_c] = mu
        rho_sum = ydual[obs0:obs0 + M * R_s].reshape(M, R_s).sum(axis=1)
        ydual[n_c + 1:n_c + 1 + M] = mu - kappa * rho_sum
        ydual[obs0 + M * R_s] = mult["slack"]
    b0 = A.shape[0] - nU
    for idx, sgn, lam in mult["bounds"]:
        ydual[b0 + idx] += sgn * lam
    stat = P @ z + q + A.T @ ydual
    F = A[:n_c, :nU].toarray()
    nu = np.linalg.lstsq(F.T, -stat[:nU], rcond=None)[0]
    ydual[:n_c] = nu
    stat = P @ z + q + A.T @ ydual
    Az = A @ z
    upper_gap, lower_gap = u - Az, Az - l
    ineq = np.arange(A.shape[0]) >= n_c
    comp = np.where(ydual > 0, ydual * np.where(np.isfinite(upper_gap), upper_gap, 0.0),
                    -ydual * np.where(np.isfinite(lower_gap), lower_gap, 0.0))
    wrong_sign = max(np.max(-ydual[ineq & ~np.isfinite(l)], initial=0.0),          # rows with only an upper bound: y >= 0
                     0.0)
    return {"primal": float(max(np.max(-upper_gap, initial=0.0), np.max(-lower_gap, initial=0.0))),
            "stationarity": float(np.max(np.abs(stat))),
            "dual_sign": float(wrong_sign),
            "complementarity": float(np.max(np.abs(comp[ineq]), initial=0.0)),
            "multiplier_scale": float(np.max(np.abs(ydual), initial=0.0)), "z": z, "y": ydual}
