"""Prototype: the CVaR constraint disaggregated -- t explicit in the master, the samples in NG groups with one epigraph
variable and one cut per group and query -- against the aggregated Kelley cut of the product path: how many oracle ROUND
TRIPS does a subproblem need?  CPU, dense fp64 oracle, small M.     python tools/proto_multicut.py M S iters NG [NG ...]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import drone as od
from tests import _host_cuts as hc
from riskaversetrajopt_amd import dense_qp

M = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
NGS = [int(a) for a in sys.argv[4:]] or [1, 4, 16]
EPS = 1e-9
rng = np.random.RandomState(7)
DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
om = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
mdl = hc.DroneReducedOracle(om)
cs = mdl.cs
nU, n = cs.nU, cs.nU + 1
aM = cs.alphaM

def rows_at(G, base, x):
    rows = G @ x + base
    arg = rows.argmax(axis=1)
    idx = np.arange(rows.shape[0])
    return rows[idx, arg], G[idx, arg]

def qp(Q, qv, F, f, A, b, nvar, bounds_idx):
    while True:
        AA = list(A); bb = list(b)
        for (i, sgn) in bounds_idx:
            e = np.zeros(nvar); e[i] = sgn; AA.append(e); bb.append(cs.u_max)
        z, lam = dense_qp.solve(Q, qv, F, f, np.array(AA).reshape(-1, nvar), np.array(bb))
        new = [(i, 1.0) for i in range(nU) if z[i] > cs.u_max + 1e-9 and (i, 1.0) not in bounds_idx] + \
              [(i, -1.0) for i in range(nU) if z[i] < cs.u_min - 1e-9 and (i, -1.0) not in bounds_idx]
        if not new:
            return z, lam
        bounds_idx += new

def cvar(m):
    w, t = cs._weights(m)
    return float(w @ m) / aM, t, w

us = mdl.initial_guess_us_mat()
for k in range(iters):
    if k < 2:
        us, _, info = mdl.solve_reduced(us, k, tol=1e-10)
        continue
    fdu, frhs, gdu, gup = mdl.linearization(us)
    G = gdu.reshape(M, -1, nU); gupf = gup.reshape(M, -1)
    uk = np.asarray(us, dtype=np.float64).reshape(-1)
    base = -(gupf - G @ uk)
    # ---- aggregated Kelley from scratch (no recycled cuts: same footing as the multi-cut loops below)
    F1 = np.hstack([fdu, np.zeros((fdu.shape[0], 1))])
    A = [np.concatenate([np.zeros(nU), [-1.0]])]; b = [0.0]; bi = []
    trips = 0
    while True:
        z, lam = qp(cs._Pd, cs.q, F1, frhs, A, b, n, bi)
        m, gi = rows_at(G, base, z[:nU] - uk)
        phi, t, w = cvar(m)
        trips += 1
        viol = phi - cs.c_s * z[nU] - cs.rhs0
        if viol <= 1e-9 or trips > 400:
            break
        g = (w[:, None] * gi).sum(0) / aM
        A.append(np.concatenate([g, [-cs.c_s]])); b.append(cs.rhs0 + (g @ z[:nU] - phi))
    z_ref, trips_ref = z.copy(), trips
    out = [f"scp {k}: Kelley {trips_ref:3d} trips"]
    # ---- multi-cut: variables (u, s, t, y_1..y_NG)
    for NG in NGS:
        if NG < 0:
            # groups by the RANK of m at the linearization point (descending): a deep-tail group that stays active, thin bands
            # around the threshold where the samples enter and leave the tail, and the rest
            edges = {-3: [0.09, 0.11, 1.0], -4: [0.085, 0.10, 0.115, 1.0], -6: [0.07, 0.09, 0.10, 0.11, 0.13, 1.0],
                     -8: [0.06, 0.08, 0.09, 0.10, 0.11, 0.12, 0.14, 1.0]}[NG]
            m_lin = rows_at(G, base, np.zeros(nU))[0]
            rank = np.empty(M, dtype=np.int64); rank[np.argsort(-m_lin, kind="stable")] = np.arange(M)
            grp = np.searchsorted(np.array(edges) * M, rank, side="right")
            label, NG = NG, len(edges)
        else:
            label = NG
            grp = (np.arange(M) * NG) // M                  # contiguous groups of samples (the device: blocks of 256)
        nv = nU + 2 + NG
        Q = np.zeros((nv, nv)); Q[:n, :n] = cs._Pd
        Q[n:, n:] = EPS * np.eye(1 + NG)
        qv = np.concatenate([cs.q, np.zeros(1 + NG)])
        F = np.hstack([fdu, np.zeros((fdu.shape[0], 2 + NG))])
        A = []; b = []
        e = np.zeros(nv); e[nU] = -1.0; A.append(e); b.append(0.0)                      # -s <= 0
        for gq in range(NG):
            e = np.zeros(nv); e[n + 1 + gq] = -1.0; A.append(e); b.append(0.0)          # -y_g <= 0
        e = np.zeros(nv); e[n] = 1.0; e[n + 1:] = 1.0 / aM; e[nU] = -cs.c_s; A.append(e); b.append(cs.rhs0)   # t + sum y / aM - c_s s <= rhs0
        bi = []
        def add_cuts(u, t_hat):
            m, gi = rows_at(G, base, u - uk)
            act = m > t_hat
            for gq in range(NG):
                sel = act & (grp == gq)
                if not sel.any():
                    continue
                Y = float((m[sel] - t_hat).sum()); gY = gi[sel].sum(0); ng = float(sel.sum())
                # y_g >= Y + gY.(u' - u) - ng (t' - t_hat)   ->   gY.u' - ng t' - y_g <= gY.u - ng t_hat - Y
                e = np.zeros(nv); e[:nU] = gY; e[n] = -ng; e[n + 1 + gq] = -1.0
                A.append(e); b.append(float(gY @ u) - ng * t_hat - Y)
            return m
        m0 = add_cuts(uk, cvar(rows_at(G, base, np.zeros(nU))[0])[1])
        trips = 1
        while True:
            z = qp(Q, qv, F, frhs, A, b, nv, bi)[0]
            u, s, t = z[:nU], z[nU], z[n]
            m = add_cuts(u, t)
            trips += 1
            phi = cvar(m)[0]
            viol = phi - cs.c_s * s - cs.rhs0
            if viol <= 1e-9 or trips > 400:
                break
        out.append(f"NG={label}: {trips:3d} trips |dz| {np.abs(z[:n] - z_ref).max():.1e}")
    print(" | ".join(out), flush=True)
    us = z_ref[:nU].reshape(S, 3)
