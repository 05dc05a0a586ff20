"""Prototype: two oracle evaluations per round trip (the master's point + a second point) -- how many ROUND TRIPS does a
subproblem need?  CPU, fp64 streaming oracle."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import drone as od
from tests import _host_cuts as hc
from riskaversetrajopt_amd import scp, dense_qp

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
MODE = sys.argv[3] if len(sys.argv) > 3 else "mid"
BETA = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
S = 50
rng = np.random.RandomState(7)
DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
om = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
mdl = hc.DroneStreamingOracle(om, nthreads=8)
cs = mdl.cs
nU, n = cs.nU, cs.nU + 1

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

us = mdl.initial_guess_us_mat()
tot1 = tot2 = 0
for k in range(iters):
    if k < 2:
        us, _, info = mdl.solve_reduced(us, k, tol=1e-9)
        continue
    fdu, frhs = mdl.final_rows(us)
    cs.us_k = np.asarray(us, dtype=np.float64).copy(); cs.u_lin = None
    F = np.hstack([fdu, np.zeros((fdu.shape[0], 1))]); f = frhs
    res = {}
    for two in (False, True):
        A = [np.concatenate([np.zeros(nU), [-1.0]])]; b = [0.0]
        bounds_idx = []
        zprev = None; trips = 0
        while True:
            z, lam = qp(cs._Pd, cs.q, F, f, A, b, bounds_idx)
            phi, t, g = cs.evaluate(None, None, 0, None, z[:nU], None)
            trips += 1
            viol = phi - cs.c_s * z[nU] - cs.rhs0
            if viol <= 1e-9 or trips > 300:
                break
            A.append(np.concatenate([g, [-cs.c_s]])); b.append(cs.rhs0 + (g @ z[:nU] - phi))
            if two and zprev is not None:
                if MODE == "mid":
                    z2 = BETA * z + (1 - BETA) * zprev
                elif MODE == "extra":
                    z2 = z + BETA * (z - zprev)
                phi2, t2, g2 = cs.evaluate(None, None, 0, None, z2[:nU], None)
                A.append(np.concatenate([g2, [-cs.c_s]])); b.append(cs.rhs0 + (g2 @ z2[:nU] - phi2))
            zprev = z
        res[two] = (trips, z.copy())
    tot1 += res[False][0]; tot2 += res[True][0]
    print(f"scp {k}: round trips one point {res[False][0]:3d} | two points ({MODE} {BETA}) {res[True][0]:3d}   |dz| {np.abs(res[False][1]-res[True][1]).max():.1e}", flush=True)
    us = res[False][1][:nU].reshape(S, 3)
print("total", tot1, tot2)
