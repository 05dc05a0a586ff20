"""Matrix-free KKT certificate of a reduced SCP subproblem against the reference's FULL QP, at any M.

``Model.solve_reduced`` solves the reference's subproblem (drone_risk.py:327-368, :413-469; driving.py:330-373, :411-456;
layout: SURVEY.md appendix A) after eliminating y_i and t exactly (cvar_cuts.py).  At M = 1e5 the full QP has 1.5e7 rows
and 7.4e8 nonzeros: no host can assemble it, let alone solve it.  But its KKT conditions at the LIFTED point

    z = (u*, y, slack, t),      t = t_risk,      y_i = max(-slack, m_i(u*) - t),       m_i(u) = max_r (G_i u - g_up_i)_r

with the master's multipliers spread over its rows

    cut k (multiplier lam_k, tail weights w_ki, rows r_ki):     kappa rho_{i, r_ki} += lam_k w_ki / (alpha M)
    CVaR row:  mu = Lam / (alpha M),  Lam = sum_k lam_k;        -y_i - slack rows:  pi_i = mu - kappa sum_r rho_ir
    -slack row: sigma;   control bounds: the master's own;      final rows: nu (least squares on the u-stationarity)

only involve SUMS OVER THE SAMPLES of quantities the device oracle holds or forms anyway: the m values at u* (one rowmax
call), the cuts' tail-row sums (the batched tail-row kernel: their gradients and offsets, recomputed -- not taken from
the master), and per cut  A_k = sum_i w_ki (m_i* - v)^+,  W_k = sum_i w_ki,  B = sum_i (m_i* - v)^+,  v = t - slack
(rato_kkt_sums).  Every complementarity product of the full QP is non-negative term by term, so the aggregated sums
below are the L1 norms of the per-row products -- an upper bound of the max-norm the host certificate of
tests/_host_cuts.py reports at M <= 1000, where the two are compared (tests/test_gpu_scp.py).

Rows are scaled to unit largest coefficient where that matters (the CVaR row carries M alpha on t: its residual is
divided by alpha M, which makes it the violation of the reduced constraint CVaR - c_s slack <= 0); dual residuals are
relative to the master-level multiplier scale max(1, |nu|, lam_k, sigma, bound multipliers).
"""
import numpy as np
import torch

from . import _lib, stats


def certify(cs, info, final_du, final_rhs, kappa=1.0):
    """``cs``: the CvarCutSolver that produced ``info`` (table-free oracle, one GPU, mode 'saa', an iteration with the
    CVaR rows); ``final_du`` (n_c, nU), ``final_rhs`` (n_c,): the equality rows of that solve.  ``kappa``: the
    reference's multiplier on the obstacle rows (0.01 drone, 1 driving) -- it cancels everywhere but the multiplier
    scale.  -> dict of residuals (0 at the optimum of the full QP)."""
    if cs.rollout is None or cs.world != 1 or cs.mode != 'saa' or cs.u_lin is None:
        raise ValueError("certify: table-free oracle on one GPU, method 'saa', delta form")
    mult = info["multipliers"]
    if mult["uncertified_cuts"]:
        raise ValueError("certify: a cut of this solve was evaluated in the scratch slot (ring exhausted)")
    nU, S, n_u, M = cs.nU, cs.S, cs.n_u, cs.M
    aM = cs.alphaM
    u = np.asarray(info["us"], dtype=np.float64).reshape(-1)
    s, t = float(info["slack"]), float(info["t_risk"])
    v = t - s
    active = [(int(sl), float(la)) for sl, la in mult["cuts"] if la > 0.0]
    K = len(active)
    lam = np.array([la for _, la in active])
    Lam = float(lam.sum())
    F = np.asarray(final_du, dtype=np.float64)
    f = np.asarray(final_rhs, dtype=np.float64)

    # (1) the m values at u* (scratch slot of the ring) -- the oracle's own round trip
    phi_star, var_star, _ = cs.evaluate(None, None, 0, None, u, slot=None)
    m_star = cs.ring_m[cs.cap - 1]

    # (2) the active cuts recomputed under the current linearization: gradient rows_k and value at u*
    rows, cut_val = np.zeros((K, nU)), np.zeros(K)
    saved = (cs.keep, cs._relin_pending)
    try:
        for c0 in range(0, K, cs.keep_max):
            chunk = [sl for sl, _ in active[c0:c0 + cs.keep_max]]
            cs.keep, cs._relin_pending = chunk, None
            r, rhs = cs.relinearize_kept_cuts(None, None, 0, None)     # cut:  r.u - c_s s <= rhs,  value(u) = r.u - (rhs - rhs0)
            rows[c0:c0 + len(chunk)] = r
            cut_val[c0:c0 + len(chunk)] = r @ u - (rhs - cs.rhs0)
    finally:
        cs.keep, cs._relin_pending = saved

    # (3) the sample sums
    nblk = (M + 255) // 256
    part = torch.zeros((nblk, 2 * K + 2), dtype=torch.float64, device=cs.device)
    slots_d = torch.as_tensor(np.array([sl for sl, _ in active] or [0], dtype=np.int32), device=cs.device)
    lam_d = torch.as_tensor(lam if K else np.zeros(1), dtype=torch.float64, device=cs.device)
    _lib.check(cs.lib.rato_kkt_sums(_lib.ptr(m_star), M, _lib.ptr(cs.ring_m), _lib.ptr(cs.ring_res), cs.nres,
                                    _lib.ptr(slots_d), _lib.ptr(lam_d), K, float(aM), float(v), _lib.ptr(part),
                                    _lib.current_stream()), "rato_kkt_sums")
    ph = part.cpu().numpy()
    A, W, B = ph[:, :K].sum(0), ph[:, K:2 * K].sum(0), float(ph[:, 2 * K].sum())
    lw_max = float(ph[:, 2 * K + 1].max())

    # ---- primal feasibility (y rows and obstacle rows hold by construction of y from the fresh m values)
    eq = float(np.max(np.abs(F @ u - f), initial=0.0))
    cvar_row = (-s * M + B) + s + aM * t                       # sum_i y_i + slack + (M alpha) t   (<= 0)
    bound = float(max(np.max(u - cs.u_max, initial=0.0), np.max(cs.u_min - u, initial=0.0), 0.0))
    primal = max(eq, max(cvar_row, 0.0) / aM, bound, max(-s, 0.0))

    # ---- stationarity
    beta = np.zeros(nU)
    for idx, sgn, la in mult["bounds"]:
        beta[np.asarray(idx, dtype=np.int64)] += sgn * np.asarray(la)
    grad_u = cs._p_diag[:nU] * u + cs.q[:nU] + lam @ rows + beta
    nu = np.linalg.lstsq(F.T, -grad_u, rcond=None)[0] if F.shape[0] else np.zeros(0)
    stat_u = float(np.max(np.abs(grad_u + F.T @ nu), initial=0.0))
    mu = Lam / aM
    sum_pi = (M * Lam - float(lam @ W)) / aM
    sigma = float(mult["slack"])
    stat_s = abs(cs._p_diag[nU] * s + cs.q[nU] + mu - sum_pi - sigma)
    stat_t = abs(Lam - float(lam @ W) / aM)                      # (M alpha) mu - kappa sum rho
    scale = max(1.0, float(np.max(np.abs(nu), initial=0.0)), float(lam.max(initial=0.0)), sigma,
                float(np.max(np.abs(beta), initial=0.0)))
    # ---- dual signs: lam_k, sigma >= 0 (NNLS), pi_i = (Lam - sum_k lam_k w_ki) / (alpha M) >= 0
    dual_sign = max(0.0, -float(lam.min(initial=0.0)), -sigma, (lw_max - Lam) / aM)
    # ---- complementarity (each a sum of non-negative products)
    comp_cvar = mu * abs(cvar_row)
    comp_y = (Lam * B - float(lam @ A)) / aM                     # sum_i pi_i (y_i + slack)
    comp_obs = float(lam @ (v + A / aM - cut_val))               # sum_ir rho_ir kappa (y_i + t - row_ir(u*))
    comp_b = float(sum(np.sum(np.asarray(la) * np.abs((cs.u_max if sgn > 0 else cs.u_min) - u[np.asarray(idx, dtype=np.int64)]))
                       for idx, sgn, la in mult["bounds"]))
    comp = max(abs(comp_cvar), abs(comp_y), abs(comp_obs), abs(sigma * s), comp_b)
    return {"primal": float(primal), "stationarity": float(max(stat_u, stat_s, stat_t) / scale),
            "dual_sign": float(dual_sign / scale), "complementarity": float(comp / scale),
            "multiplier_scale": float(scale), "full_qp_multiplier_scale": float(max(scale, Lam / (aM * kappa))),
            "active_cuts": K, "cvar_at_solution": float(phi_star), "var_at_solution": float(var_star),
            "parts": {"eq": eq, "cvar_row": float(cvar_row / aM), "stat_u": stat_u, "stat_s": float(stat_s),
                      "stat_t": float(stat_t), "comp_cvar": float(comp_cvar), "comp_y": float(comp_y),
                      "comp_obs": float(comp_obs), "comp_slack": float(abs(sigma * s)), "comp_bounds": comp_b}}
