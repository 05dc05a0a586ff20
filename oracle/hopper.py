"""Oracle (test infrastructure) — hopper uncertain-friction SAA constraint,
NumPy fp64.

Restates the sample-dependent part of ``/root/reference/hopper/hopper.py``:
the random-Fourier friction field (:68-81), the slip-risk rows (:300-367), the
slices of the IPOPT Jacobian / Lagrangian-Hessian that depend on the samples
(reference: ``jacrev(g)`` :569, ``hessian(lambda.g)`` :577-580) and the
Monte-Carlo check (:901-925).  Pinned by executing the reference's own text — see ``oracle/__init__.py``.
The sample-independent NLP rows (RK4 defects, contact equalities, bounds) are
out of scope.
"""
import numpy as np

# ---- constants: hopper.py:44-89 --------------------------------------------
S_DEFAULT, M_DEFAULT = 30, 30        # :45-46
T = 2.0                              # :47
n_x, n_u = 8, 4                      # :51-57
mu_nom = 0.10                        # :68
num_mu_features = 30                 # :69


def phase_times(S):
    """hopper.py:48-49 has time_jump=10, time_land=20 at S=30 -> S/3, 2S/3."""
    return S // 3, (2 * S) // 3


def sample_friction_fields(rng, M):
    """hopper.py:70-74 (and the MC resample :975-979): three bulk uniforms in
    this order on ``rng`` (``RandomState(1)`` == the script's ``seed(1)``)."""
    intensities = rng.uniform(0, 1, (M, num_mu_features))
    intensities = np.sqrt(2 / num_mu_features) * intensities
    intensities = 0.025 * intensities
    thetas = rng.uniform(0, np.pi, (M, num_mu_features))
    taus = rng.uniform(0, 2 * np.pi, (M, num_mu_features))
    return intensities, thetas, taus


def friction_at_px(position_x, intensities, thetas, taus):
    """hopper.py:75-81, broadcast: position_x (C,), fields (M,K) -> mu (M,C)."""
    px = np.asarray(position_x, dtype=np.float64)
    arg = thetas[:, None, :] * px[None, :, None] + taus[:, None, :]
    return mu_nom + np.sum(intensities[:, None, :] * np.cos(arg), axis=-1)


def friction_derivatives(position_x, intensities, thetas, taus):
    """mu, mu', mu'' at every (sample, contact): each (M,C)."""
    px = np.asarray(position_x, dtype=np.float64)
    arg = thetas[:, None, :] * px[None, :, None] + taus[:, None, :]
    a, th = intensities[:, None, :], thetas[:, None, :]
    mu = mu_nom + np.sum(a * np.cos(arg), axis=-1)
    dmu = -np.sum(a * th * np.sin(arg), axis=-1)
    d2mu = -np.sum(a * th * th * np.cos(arg), axis=-1)
    return mu, dmu, d2mu


class Model:
    """hopper.py:90-171, 300-367 — only what touches the sample axis."""

    def __init__(self, intensities, thetas, taus, method='saa', alpha=0.1, S=S_DEFAULT):
        self.method, self.alpha, self.S = method, alpha, S
        self.time_jump, self.time_land = phase_times(S)
        z = 0.0 if method == 'baseline' else 1.0          # :97-100 zeroes the fields
        self.intensities = z * np.asarray(intensities, dtype=np.float64)
        self.thetas = z * np.asarray(thetas, dtype=np.float64)
        self.taus = z * np.asarray(taus, dtype=np.float64)
        self.M = self.intensities.shape[0]
        self.num_vars = (S + 1) * n_x + S * n_u + self.M + 2

    # -- variable layout: hopper.py:105-132 ----------------------------------
    def convert_z_to_variables(self, z):
        S = self.S
        nx, nu = (S + 1) * n_x, S * n_u
        return z[:nx], z[nx:nx + nu], z[nx + nu:-2], z[-2], z[-1]

    def convert_z_to_xs_us_mats(self, z):
        xs_vec, us_vec, _, _, _ = self.convert_z_to_variables(z)
        return (np.reshape(xs_vec, (n_x, self.S + 1), 'F').T.copy(),
                np.reshape(us_vec, (n_u, self.S), 'F').T.copy())

    def end_effector_position(self, x):                    # :166-171
        return np.stack([x[..., 0] + x[..., 3] * np.sin(x[..., 2]),
                         x[..., 1] - x[..., 3] * np.cos(x[..., 2])], axis=-1)

    def contact_steps(self):
        """time indices of the contact phases [0,time_jump) U [time_land,S) (:306-311)."""
        return np.concatenate([np.arange(0, self.time_jump),
                               np.arange(self.time_land, self.S)])

    def end_effector_x_derivatives(self, x):
        """Chain-rule factors of the (sample-independent) map state -> end-effector x position p = x0 + x3 sin x2
        (hopper.py:166-171) that carry dh/dpx and d2h/dpx2 to the NLP variables (x0, x2, x3) in jac_g / the
        Hessian (hopper.py:569,577-580):  -> (J (...,3) = dp/d(x0,x2,x3),  H (...,3,3) = d2p/d(x0,x2,x3)^2)."""
        x = np.asarray(x, dtype=np.float64)
        s, c = np.sin(x[..., 2]), np.cos(x[..., 2])
        J = np.stack([np.ones_like(s), x[..., 3] * c, s], axis=-1)
        H = np.zeros(x.shape[:-1] + (3, 3))
        H[..., 1, 1] = -x[..., 3] * s
        H[..., 1, 2] = H[..., 2, 1] = c
        return J, H

    def contact_chain(self, Z):
        """The factors above at the contact steps of ``contact_inputs(Z)``: J (C,3), H (C,3,3).  With the device
        outputs:  dh_ic/d(x0,x2,x3)_c = dh_dpx[i,c] J[c];  sum_i lam_ic d2h_ic/d(.)2 = D2[c] J[c] J[c]' + (sum_i lam_ic
        dh_dpx[i,c]) H[c];  mixed with fz: D1[c] J[c]."""
        xs_mat, _ = self.convert_z_to_xs_us_mats(Z)
        xc = np.concatenate([xs_mat[:self.time_jump], xs_mat[self.time_land:-1]])
        return self.end_effector_x_derivatives(xc)

    def contact_inputs(self, Z):
        """(px (C,), forces (C,2)) as gathered at :305-311."""
        xs_mat, us_mat = self.convert_z_to_xs_us_mats(Z)
        ee_x = self.end_effector_position(xs_mat)[:, 0]
        px = np.concatenate([ee_x[:self.time_jump], ee_x[self.time_land:-1]])
        forces = np.concatenate([us_mat[:self.time_jump, 2:], us_mat[self.time_land:, 2:]])
        return px, forces

    def no_slip_values(self, px, forces):
        """fx - mu_i(px_c) fz  for every (sample, contact): (M,C) (:315-323)."""
        mu = friction_at_px(px, self.intensities, self.thetas, self.taus)
        return forces[None, :, 0] - mu * forces[None, :, 1]

    # -- hopper.py:300-367 ---------------------------------------------------
    def slip_risk_constraints(self, Z):
        _, _, ys, slack_var, t_risk = self.convert_z_to_variables(Z)
        px, forces = self.contact_inputs(Z)
        M, C = self.M, forces.shape[0]
        h = self.no_slip_values(px, forces)
        if self.method == 'baseline':                      # :339-348
            return (h - slack_var).reshape(M * C)
        gs = np.zeros(1 + M + M * C + 1)                   # :351 (last entry stays 0)
        gs[0] = (M * self.alpha) * t_risk + np.sum(ys)     # :354
        gs[1:1 + M] = -ys                                  # :357
        gs[1 + M:1 + M + M * C] = (h - t_risk - ys[:, None] - slack_var).reshape(M * C)  # :359-366
        return gs

    # -- sample-dependent slices of jac_g / hess (reference: autodiff) -------
    def slip_partials(self, px, forces):
        """Per (sample, contact) first derivatives of h = fx - mu(px) fz:
        dh/dfx = 1, dh/dfz = -mu, dh/dpx = -mu'(px) fz.  Returns (h, dh_dfz, dh_dpx), each (M,C)."""
        mu, dmu, _ = friction_derivatives(px, self.intensities, self.thetas, self.taus)
        fx, fz = forces[None, :, 0], forces[None, :, 1]
        return fx - mu * fz, -mu, -dmu * fz

    def slip_hessian_sums(self, px, forces, lam):
        """lambda-weighted second derivatives reduced over samples, per contact:
        D1_c = sum_i lam_ic d2h/(dpx dfz) = -sum_i lam_ic mu_i'(px_c)
        D2_c = sum_i lam_ic d2h/dpx^2     = -fz_c sum_i lam_ic mu_i''(px_c)."""
        _, dmu, d2mu = friction_derivatives(px, self.intensities, self.thetas, self.taus)
        D1 = -np.sum(lam * dmu, axis=0)
        D2 = -forces[:, 1] * np.sum(lam * d2mu, axis=0)
        return D1, D2

    # -- the reference's own matrices: jacrev(g) / hessian(lambda.g) restricted to the slip rows ---------------------
    def slip_jacobian(self, Z):
        """``jacrev(slip_risk_constraints)(Z)`` (hopper.py:569 on the rows of :300-367) as a scipy CSC matrix with the
        reference's rows (1 + M + i C + c, 'saa'; i C + c, 'baseline') and columns (Z order: xs [t][8], us [t][4], y, slack,
        t_risk); exact zeros dropped, as ``csc_matrix(dense)`` drops them."""
        import scipy.sparse as sp
        Z = np.asarray(Z, dtype=np.float64)
        M, S = self.M, self.S
        px, forces = self.contact_inputs(Z)
        C = px.shape[0]
        _, dfz, dpx = self.slip_partials(px, forces)
        Jee, _ = self.contact_chain(Z)
        steps = self.contact_steps()
        nX, nU = (S + 1) * n_x, S * n_u
        saa = self.method != 'baseline'
        r0 = 1 + M if saa else 0
        rows = (r0 + np.arange(M)[:, None] * C + np.arange(C)[None, :])              # (M,C)
        I, Jc, V = [], [], []

        def put(r, c, v):
            r, c, v = np.broadcast_arrays(r, c, v)
            I.append(r.reshape(-1)), Jc.append(c.reshape(-1)), V.append(np.asarray(v, dtype=np.float64).reshape(-1))
        for k, xk in enumerate((0, 2, 3)):                                            # d/d(x0, x2, x3) of step t_c
            put(rows, (steps * n_x + xk)[None, :], dpx * Jee[None, :, k])
        put(rows, (nX + steps * n_u + 2)[None, :], 1.0)                               # d/dfx
        put(rows, (nX + steps * n_u + 3)[None, :], dfz)                               # d/dfz = -mu
        put(rows, self.num_vars - 2, -1.0)                                            # slack (:347, :366)
        if saa:
            put(rows, (nX + nU + np.arange(M))[:, None], -1.0)                        # -y_i (:366)
            put(rows, self.num_vars - 1, -1.0)                                        # -t_risk
            put(0, nX + nU + np.arange(M), 1.0)                                       # row 0 (:354)
            put(0, self.num_vars - 1, M * self.alpha)
            put(1 + np.arange(M), nX + nU + np.arange(M), -1.0)                       # -y_i <= 0 (:357)
        n_rows = (1 + M + M * C + 1) if saa else M * C
        A = sp.coo_matrix((np.concatenate(V), (np.concatenate(I), np.concatenate(Jc))),
                          shape=(n_rows, self.num_vars)).tocsc()
        A.eliminate_zeros()
        A.sort_indices()
        return A

    def slip_hessian(self, Z, lam):
        """``hessian(lambda . slip_risk_constraints)(Z)`` (hopper.py:575-580; lam (M,C) on the per-sample rows: every other
        row is linear) as a scipy CSC (num_vars x num_vars): per contact the 3 x 3 block on (x0, x2, x3) of its step and
        the mixed entries with fz."""
        import scipy.sparse as sp
        Z = np.asarray(Z, dtype=np.float64)
        px, forces = self.contact_inputs(Z)
        _, _, dpx = self.slip_partials(px, forces)
        D1, D2 = self.slip_hessian_sums(px, forces, lam)
        D0 = np.sum(lam * dpx, axis=0)
        return hessian_from_sums(self, Z, D0, D1, D2)

    # -- Monte-Carlo validation: hopper.py:901-925 ---------------------------
    def no_slip_constraints_verification(self, px, forces):
        Z = self.no_slip_values(px, forces).max(axis=1)
        return Z <= 1e-6, Z


def hessian_from_sums(model, Z, D0, D1, D2):
    """The Lagrangian-Hessian blocks from the three per-contact sample sums D0_c = sum_i lam_ic dh_ic/dpx,
    D1_c = sum_i lam_ic d2h/(dpx dfz), D2_c = sum_i lam_ic d2h/dpx2 and the chain factors of p = x0 + x3 sin x2."""
    import scipy.sparse as sp
    S = model.S
    Jee, Hee = model.contact_chain(Z)
    steps = model.contact_steps()
    nX = (S + 1) * n_x
    H = sp.lil_matrix((model.num_vars, model.num_vars))
    for c, t in enumerate(steps):
        xi = [t * n_x + 0, t * n_x + 2, t * n_x + 3]
        blk = D2[c] * np.outer(Jee[c], Jee[c]) + D0[c] * Hee[c]
        fz = nX + t * n_u + 3
        for a in range(3):
            for b in range(3):
                H[xi[a], xi[b]] = blk[a, b]
            H[xi[a], fz] = H[fz, xi[a]] = D1[c] * Jee[c, a]
    H = H.tocsc()
    H.eliminate_zeros()
    H.sort_indices()
    return H
