"""Prototype: SQP with a sample-density Hessian of the CVaR constraint vs Kelley (CPU, dense fp64 oracle, small M)."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import drone as od
from tests import _host_cuts as hc
from riskaversetrajopt_amd import scp, dense_qp

M = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
HFRAC = float(sys.argv[4]) if len(sys.argv) > 4 else 0.02      # boundary band: this fraction of the samples on each side
rng = np.random.RandomState(7)
DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
om = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
mdl = hc.DroneReducedOracle(om)
cs = mdl.cs
nU, n = cs.nU, cs.nU + 1
aM = cs.alphaM

def oracle(G, base, x, want_H=True):
    rows = G @ x + base
    arg = rows.argmax(axis=1)
    idx = np.arange(rows.shape[0])
    m = rows[idx, arg]
    w, t = cs._weights(m)
    gi = G[idx, arg]
    g = (w[:, None] * gi).sum(0) / aM
    phi = float(w @ m) / aM
    H = None
    if want_H:
        order = np.sort(m)
        k = int(M - np.floor(aM) - 1)
        nb = max(int(HFRAC * M), 8)
        lo, hi = order[max(k - nb, 0)], order[min(k + nb, M - 1)]
        B = (m >= lo) & (m <= hi)
        width = hi - lo
        gb = gi[B]
        d = gb - gb.mean(0)
        H = d.T @ d / (aM * width)               # density (|B| / width) x covariance of the gradients in the band / (alpha M) ... 
        # row switching inside tail samples
        rows2 = rows.copy(); rows2[idx, arg] = -np.inf
        arg2 = rows2.argmax(axis=1); m2 = rows2[idx, arg2]
        hs = width
        sw = (w > 0) & (m - m2 < hs)
        if sw.any():
            dd = gi[sw] - G[idx[sw], arg2[sw]]
            H = H + (w[sw, None] * dd).T @ dd / (aM * 2 * hs)
    return phi, t, g, H

def qp(Q, qv, F, f, A, b, bounds_idx):
    while True:
        AA = list(A); bb = list(b)
        for (i, sgn) in bounds_idx:
            e = np.zeros(n); e[i] = sgn; AA.append(e); bb.append(cs.u_max)
        z, lam = dense_qp.solve(Q, qv, F, f, np.array(AA).reshape(-1, n), np.array(bb))
        new = [(i, 1.0) for i in range(nU) if z[i] > cs.u_max + 1e-9 and (i, 1.0) not in bounds_idx] + \
              [(i, -1.0) for i in range(nU) if z[i] < cs.u_min - 1e-9 and (i, -1.0) not in bounds_idx]
        if not new:
            return z, lam
        bounds_idx += new


SWITCH = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-2
us = mdl.initial_guess_us_mat()
tot_k = tot_h = 0
for k in range(iters):
    if k < 2:
        us, _, info = mdl.solve_reduced(us, k, tol=1e-10)
        continue
    fdu, frhs, gdu, gup = mdl.linearization(us)
    G = gdu.reshape(M, -1, nU); gupf = gup.reshape(M, -1)
    uk = np.asarray(us, dtype=np.float64).reshape(-1)
    base = -(gupf - G @ uk)
    F = np.hstack([fdu, np.zeros((fdu.shape[0], 1))]); f = frhs
    cs.keep = []                                    # (no recycling in either leg: compare the loops themselves)
    us_ref, _, info = mdl.solve_reduced(us, k, tol=1e-10)
    z_ref = np.concatenate([us_ref.reshape(-1), [info["slack"]]])
    Pd, q = cs._Pd, cs.q
    A = [np.concatenate([np.zeros(nU), [-1.0]])]; b = [0.0]
    bounds_idx = []
    hist = []
    newton = False
    lam_c = 0.0; H = None; zc = None
    for it in range(200):
        if newton and H is not None:
            Q = Pd.copy(); Q[:nU, :nU] += lam_c * H
            qq = q.copy(); qq[:nU] -= lam_c * (H @ zc[:nU])
            z, lam = qp(Q, qq, F, f, A, b, bounds_idx)
        else:
            z, lam = qp(Pd, q, F, f, A, b, bounds_idx)
        lam_c = float(sum(lam[1:len(A)]))
        x = z[:nU] - uk
        phi, t, g, Hn = oracle(G, base, x, want_H=True)
        viol = phi - cs.c_s * z[nU] - cs.rhs0
        hist.append(("N" if newton else "K", np.abs(z - z_ref).max(), viol))
        if newton and zc is not None and np.abs(z - zc).max() < 1e-8 and abs(viol) < 1e-9:
            break
        if not newton and viol <= 1e-10:
            break
        A.append(np.concatenate([g, [-cs.c_s]])); b.append(cs.rhs0 + (g @ z[:nU] - phi))
        if viol <= SWITCH:
            newton = True
        H, zc = Hn, z.copy()
    tot_k += info["cuts"]; tot_h += len(hist)
    print(f"scp {k}: Kelley cuts {info['cuts']:3d} | hybrid queries {len(hist):3d}: " + " ".join("%s%.0e/%.0e" % h for h in hist[-14:]), flush=True)
    us = us_ref
print("total Kelley", tot_k, "hybrid", tot_h)
