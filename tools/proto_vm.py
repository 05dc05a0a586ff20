"""Prototype: variable-metric stabilised query points, Kelley-certified answer (CPU, fp64 streaming oracle)."""
import sys, time, math, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import drone as od
from tests import _host_cuts as hc
from riskaversetrajopt_amd import scp, dense_qp

MODE = sys.argv[3] if len(sys.argv) > 3 else "vm"
CARRY = int(sys.argv[4]) if len(sys.argv) > 4 else 1
DELTA = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-7

class VM:
    def __init__(self, cs):
        self.cs = cs
        self.B = None
    def qp(self, Q, q, F, f, rows, rhs, bounds_idx):
        cs = self.cs; nU = cs.nU; n = nU + 1
        while True:
            A = [r for r in rows]; b = list(rhs)
            for (i, sgn) in bounds_idx:
                e = np.zeros(n); e[i] = sgn; A.append(e); b.append(cs.u_max)
            z, lam = dense_qp.solve(Q, q, F, f, np.array(A).reshape(-1, n), np.array(b))
            new = [(i, 1.0) for i in range(nU) if z[i] > cs.u_max + 1e-9 and (i, 1.0) not in bounds_idx] + \
                  [(i, -1.0) for i in range(nU) if z[i] < cs.u_min - 1e-9 and (i, -1.0) not in bounds_idx]
            if not new:
                return z, lam
            bounds_idx += new
    def solve(self, final_du, final_rhs, tol=1e-9, max_cuts=400, verbose=False):
        cs = self.cs; nU = cs.nU; n = nU + 1
        F = np.hstack([final_du, np.zeros((final_du.shape[0], 1))]); f = final_rhs
        Pd = cs._Pd; q = cs.q
        rows = [np.concatenate([np.zeros(nU), [-1.0]])]; rhs = [0.0]     # -slack <= 0
        cut_rows = []
        kept = list(cs.keep)
        if kept:
            R, r = cs.relinearize_kept_cuts(None, None, 0, None)
            for k, sl in enumerate(kept):
                rows.append(np.concatenate([R[k], [-cs.c_s]])); rhs.append(r[k]); cut_rows.append((len(rows) - 1, sl))
        free = [sl for sl in range(cs.cap - 1) if sl not in set(kept)]
        bounds_idx = []
        n_q = 0; n_plain = 0
        uq_prev = g_prev = None
        if not CARRY: self.B = None
        status = "solved"
        while True:
            zb, lam = self.qp(Pd, q, F, f, rows, rhs, bounds_idx)
            lam_c = sum(lam[r] for r, _ in cut_rows if r < len(lam))
            zq = zb; plain = True
            if MODE == "vm" and self.B is not None and uq_prev is not None and lam_c > 0:
                Q = Pd.copy(); Q[:nU, :nU] += lam_c * self.B
                qq = q.copy(); qq[:nU] -= lam_c * (self.B @ uq_prev)
                zs, _ = self.qp(Q, qq, F, f, rows, rhs, bounds_idx)
                if np.abs(zs - zb).max() > DELTA:
                    zq = zs; plain = False
            slot = free.pop() if free else None
            phi, tstar, g = cs.evaluate(None, None, 0, None, zq[:nU], slot)
            n_q += 1; n_plain += plain
            viol = phi - cs.c_s * zq[nU] - cs.rhs0
            if verbose:
                print(f"    q{n_q:3d} {'K' if plain else 'N'} viol {viol:+.3e} |zq-zb| {np.abs(zq-zb).max():.2e} lam {lam_c:.3e}")
            if uq_prev is not None:
                s = zq[:nU] - uq_prev; y = g - g_prev
                sy = s @ y
                if sy > 1e-10 * np.linalg.norm(s) * np.linalg.norm(y) and np.linalg.norm(s) > 1e-9:
                    if self.B is None:
                        self.B = (y @ y / sy) * np.eye(nU)
                    Bs = self.B @ s
                    self.B = self.B - np.outer(Bs, Bs) / (s @ Bs) + np.outer(y, y) / sy
            uq_prev, g_prev = zq[:nU].copy(), g.copy()
            if plain and viol <= tol:
                if viol > 1e-11:
                    rows.append(np.concatenate([g, [-cs.c_s]])); rhs.append(cs.rhs0 + (g @ zq[:nU] - phi))
                    if slot is not None: cut_rows.append((len(rows) - 1, slot))
                    zb, lam = self.qp(Pd, q, F, f, rows, rhs, bounds_idx)
                    zq = zb
                break
            if n_q > max_cuts:
                status = "max"; break
            rows.append(np.concatenate([g, [-cs.c_s]])); rhs.append(cs.rhs0 + (g @ zq[:nU] - phi))
            if slot is not None: cut_rows.append((len(rows) - 1, slot))
        # keep rule (as cvar_cuts)
        for row, sl in cut_rows:
            active = row < lam.shape[0] and lam[row] > 1e-12
            cs.idle[sl] = 0 if active else cs.idle.get(sl, 0) + 1
        act = [sl for row, sl in reversed(cut_rows) if cs.idle[sl] <= cs.keep_idle]
        recent = [sl for _, sl in reversed(cut_rows)][:cs.keep_recent]
        keep = []
        for sl in act + recent:
            if sl not in keep: keep.append(sl)
        cs.keep = keep[:cs.keep_max]; cs.idle = {sl: cs.idle[sl] for sl in cs.keep}
        return zq[:nU].reshape(cs.S, cs.n_u).copy(), n_q, n_plain, status

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S = 50
rng = np.random.RandomState(7)
DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
om = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
mdl = hc.DroneStreamingOracle(om, nthreads=8)
vm = VM(mdl.cs)
us = mdl.initial_guess_us_mat()
tot = 0
ref = None
for k in range(iters):
    t0 = time.time()
    if k < 2:
        us_new, _, info = mdl.solve_reduced(us, k, tol=1e-9); nq = npl = 0; st = info["status"]
    else:
        fdu, frhs = mdl.final_rows(us)
        mdl.cs.us_k = np.asarray(us, dtype=np.float64).copy()
        mdl.cs.u_lin = None
        us_new, nq, npl, st = vm.solve(fdu, frhs, verbose=(k == 2 and len(sys.argv) > 6))
    err = scp.L2_error_us(us_new, us)
    tot += nq
    print(f"scp {k:2d} queries {nq:3d} (plain {npl:3d}) {st} L2 {err:.6e} {time.time()-t0:.2f}s", flush=True)
    us = us_new
print("total queries", tot)
np.save(f"/tmp/us_{MODE}_{CARRY}.npy", us)
