"""Sparse QP assembly — the reference's L3 layer without its O(M^2) dense matrix.

The reference packs the per-sample linearizations into a DENSE
``(1 + M + M*n_obs*S + 1 + n_x) x (n_u*S + M + 2)`` matrix (48 GB at M = 1e4),
converts it with ``sp.csr_matrix(dense)`` (which drops exact zeros) and stacks
the control bounds underneath (``drone_risk.py:282-423``, ``driving.py:301-421``).
Here the same ``(A csc, l, u)`` — same row order, same column order, same
dropped-zero pattern, same ``scp_iter`` relaxations — is built directly in
sparse form from the kernels' packed outputs.

Variable vector  z = (u[0:n_u*S], y[0:M], slack, t_risk)   (drone_risk.py:329-333,460-461)
Rows of A (SAA):  n_c final rows | CVaR sum row | M rows (-y_i - slack <= 0) |
                  M*R_s linearized constraint rows (sample-major, then j, then t) |
                  -slack <= 0 | n_u*S control-bound rows.
"""
import numpy as np
import scipy.sparse as sp


def _pair_index(S):
    """(t, s) of every packed pair, in pair order (row-major in t)."""
    t = np.concatenate([np.full(tt, tt, dtype=np.int64) for tt in range(1, S)]) if S > 1 else np.zeros(0, np.int64)
    s = np.concatenate([np.arange(tt, dtype=np.int64) for tt in range(1, S)]) if S > 1 else np.zeros(0, np.int64)
    return t, s


def _finish(rows, cols, vals, shape):
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    keep = vals != 0.0                      # sp.csr_matrix(dense) drops exact zeros (drone_risk.py:419)
    A = sp.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=shape).tocsc()
    A.sort_indices()
    return A


def saa_constraints(final_du, final_rhs, G_packed, g_up, *, n_u, S, M, alpha, method, kappa,
                    baseline_pad, u_min, u_max, relax):
    """Generic assembler.

    final_du  (n_c, n_u*S)    sample mean of the final-constraint Jacobian
    final_rhs (n_c,)          sample mean of -v_final + v_final_du.u  (lower == upper)
    G_packed  (n_pairs, n_g, R, M) packed causal Jacobian: entry [pair(t,s), g, r, i] is
              d row(r, t) / d u[s, ctrl_of_g]; rows of sample i are ordered r*S + t
              (drone: g = axis in {0,1}, r = obstacle; car: g = control, R = 1)
    g_up      (R, S, M)
    relax     None | ('scale', first_row, factor, lo, hi) | ('zero', first_row)
    -> (A csc, l, u)
    """
    final_du = np.asarray(final_du, dtype=np.float64)
    final_rhs = np.asarray(final_rhs, dtype=np.float64)
    G_packed = np.asarray(G_packed, dtype=np.float64)
    g_up = np.asarray(g_up, dtype=np.float64)
    n_c = final_du.shape[0]
    n_g, R = G_packed.shape[1], G_packed.shape[2]
    R_s = R * S
    nU = n_u * S
    ncols = nU + M + 2
    col_slack, col_t = nU + M, nU + M + 1
    saa = method == 'saa'
    n_head = (1 + M) if saa else 0
    n_obs_rows = n_head + M * R_s + (1 if saa else 0)
    n_rows_As = n_c + n_obs_rows
    rows, cols, vals = [], [], []

    # final rows
    rr, cc = np.nonzero(final_du)
    rows.append(rr.astype(np.int64)); cols.append(cc.astype(np.int64)); vals.append(final_du[rr, cc])
    low = np.full(n_rows_As, -np.inf)
    up = np.full(n_rows_As, np.inf)
    low[:n_c] = final_rhs
    up[:n_c] = final_rhs

    base = n_c
    if saa:
        # (M alpha) t + sum_i y_i (+ slack: the reference's slice (n_u*S):-1 includes it) <= 0
        rows.append(np.full(M + 2, base, dtype=np.int64))
        cols.append(np.concatenate([np.arange(nU, nU + M + 1), [col_t]]).astype(np.int64))
        vals.append(np.concatenate([np.ones(M + 1), [M * alpha]]))
        up[base] = 0.0
        # -y_i - slack <= 0
        i = np.arange(M, dtype=np.int64)
        rows.append(np.concatenate([base + 1 + i, base + 1 + i]))
        cols.append(np.concatenate([nU + i, np.full(M, col_slack, dtype=np.int64)]))
        vals.append(-np.ones(2 * M))
        up[base + 1:base + 1 + M] = 0.0
    obs0 = base + n_head

    # linearized constraint rows
    tt, ss = _pair_index(S)
    if tt.size:
        i = np.arange(M, dtype=np.int64)
        for g in range(n_g):
            for r in range(R):
                blk = G_packed[:, g, r, :]                                   # (n_pairs, M)
                rws = obs0 + i[None, :] * R_s + r * S + tt[:, None]          # (n_pairs, M)
                cls = np.broadcast_to((ss * n_u + g)[:, None], blk.shape)
                rows.append(rws.reshape(-1)); cols.append(cls.reshape(-1).astype(np.int64))
                vals.append((kappa * blk).reshape(-1))
    all_rows = obs0 + np.arange(M * R_s, dtype=np.int64)
    up[obs0:obs0 + M * R_s] = kappa * np.transpose(g_up, (2, 0, 1)).reshape(-1) - baseline_pad
    if saa:
        yi = nU + np.repeat(np.arange(M, dtype=np.int64), R_s)
        rows.append(np.concatenate([all_rows, all_rows]))
        cols.append(np.concatenate([yi, np.full(M * R_s, col_t, dtype=np.int64)]))
        vals.append(np.full(2 * M * R_s, -kappa))
        last = obs0 + M * R_s
        rows.append(np.array([last], dtype=np.int64)); cols.append(np.array([col_slack], dtype=np.int64))
        vals.append(np.array([-1.0]))
        up[last] = 0.0

    # scp_iter relaxations act on the dense block BEFORE the control bounds are stacked
    if relax is not None:
        first = relax[1]
        r_all, v_all = np.concatenate(rows), np.concatenate(vals)
        sel = r_all >= first
        if relax[0] == 'scale':            # drone_risk.py:413-417
            v_all = np.where(sel, v_all * relax[2], v_all)
            low[first:] = relax[3]
            up[first:] = relax[4]
        else:                              # driving.py:411-415 (entries become 0 and are dropped)
            v_all = np.where(sel, 0.0, v_all)
            low[first:] = 0.0
            up[first:] = 0.0
        c_all = np.concatenate(cols)
        rows, cols, vals = [r_all], [c_all], [v_all]

    # control bounds (identity on the u columns)
    k = np.arange(nU, dtype=np.int64)
    rows.append(n_rows_As + k); cols.append(k); vals.append(np.ones(nU))
    l = np.concatenate([low, np.full(nU, float(u_min))])
    u = np.concatenate([up, np.full(nU, float(u_max))])
    A = _finish(rows, cols, vals, (n_rows_As + nU, ncols))
    return A, l, u


def objective(n_u, S, M, dt, R, slack_penalty):
    """drone_risk.py:376-391 / driving.py:375-389:  P = blockdiag(2 dt R) on u, slack penalty on both
    P[slack,slack] and q[slack]."""
    n = n_u * S + M + 2
    blocks = sp.kron(sp.eye(S), sp.csc_matrix(2.0 * dt * np.asarray(R, dtype=np.float64)))
    P = sp.lil_matrix((n, n))
    P[:n_u * S, :n_u * S] = blocks
    P[n - 2, n - 2] = slack_penalty
    q = np.zeros(n)
    q[n - 2] = slack_penalty
    P = P.tocsc()
    P.eliminate_zeros()
    return P, q


class FastAssembler:
    """Per-iteration assembly with a cached sparsity pattern and the device-emitted value block.

    The pattern of A is iteration invariant (the reference relies on it: ``update(Ax=A.data)``,
    drone_risk.py:451), so it is taken once from a host-assembled matrix; afterwards an iteration only
    moves the nnz(G) values that ``rato_emit_csc_values`` already wrote in CSC order, the few
    final-constraint entries and the bounds.  Entries that are structurally present but happen to be
    exactly 0.0 in a later iterate stay as explicit zeros (the reference would drop them; OSQP does not
    care).  If the first matrix lacks a structural entry (an exact zero was dropped) the fast path is
    declined and the caller keeps using the host assembler.
    """

    def __init__(self, A0, l0, u0, *, n_c, n_u, n_g, R, S, M, saa):
        A0 = A0.tocsc()
        A0.sort_indices()
        self.shape = A0.shape
        self.indptr, self.indices = A0.indptr.copy(), A0.indices.copy()
        self.base = A0.data.copy()                       # constants (CVaR rows, identity); rest overwritten
        self.l0, self.u0 = l0.copy(), u0.copy()
        self.n_c, self.S, self.M, self.R, self.n_u, self.n_g = n_c, S, M, R, n_u, n_g
        R_s = R * S
        n_head = (1 + M) if saa else 0
        self.obs0 = n_c + n_head
        self.obs1 = self.obs0 + M * R_s
        self.ok = True
        # obstacle block: data positions in emission order (column (s,g) -> [i][r][t>s])
        pos = []
        for s in range(S - 1):
            for g in range(n_g):
                c = s * n_u + g
                lo, hi = self.indptr[c], self.indptr[c + 1]
                rows = self.indices[lo:hi]
                a, b = lo + np.searchsorted(rows, self.obs0), lo + np.searchsorted(rows, self.obs1)
                if b - a != M * R * (S - 1 - s):
                    self.ok = False
                    return
                pos.append(np.arange(a, b, dtype=np.int64))
        self.G_pos = np.concatenate(pos) if pos else np.zeros(0, np.int64)
        # final-constraint entries
        self.fin_pos = np.nonzero(self.indices < n_c)[0]
        self.fin_row = self.indices[self.fin_pos]
        self.fin_col = np.searchsorted(self.indptr, self.fin_pos, side='right') - 1

    def assemble(self, G_vals, final_du, final_rhs, g_up_irt, *, kappa, baseline_pad, relax):
        """G_vals: device-emitted values (already multiplied by kappa * relax factor), host ndarray;
        g_up_irt: (M, R, S) host array.  -> (A csc, l, u)."""
        data = self.base.copy()
        l, u = self.l0.copy(), self.u0.copy()
        data[self.G_pos] = G_vals
        data[self.fin_pos] = np.asarray(final_du, dtype=np.float64)[self.fin_row, self.fin_col]
        l[:self.n_c] = final_rhs
        u[:self.n_c] = final_rhs
        u[self.obs0:self.obs1] = kappa * np.asarray(g_up_irt, dtype=np.float64).reshape(-1) - baseline_pad
        if relax is not None:                      # ('scale', first_row, factor, lo, hi): drone_risk.py:413-417
            first, factor = relax[1], relax[2]
            n_As = self.shape[0] - self.n_u * self.S
            sel = (self.indices >= first) & (self.indices < n_As)
            sel[self.G_pos] = False                # the emitted block already carries the factor
            data[sel] *= factor
            l[first:n_As] = relax[3]
            u[first:n_As] = relax[4]
        A = sp.csc_matrix((data, self.indices, self.indptr), shape=self.shape)
        return A, l, u
