"""Oracle (test infrastructure) — drone SAA inner loop, NumPy fp64.

Restates, vectorised over the sample axis, the arithmetic of
``/root/reference/drone/drone_risk.py`` (+ ``drone_params.py``,
``drone_utils.py``).  Pinned by executing the reference's own text — see ``oracle/__init__.py``.

The reference differentiates with ``jax.jacfwd`` (drone_risk.py:255-258); here
the control-Jacobian is the analytic forward-sensitivity recursion of the same
Euler–Maruyama map, which is what forward-mode autodiff computes.  The
recursion is checked against independent autodiff and finite differences in
``tests/test_oracle_drone.py``.
"""
import numpy as np

# ---- constants: drone_params.py:1-45 ---------------------------------------
OSQP_TOL = 1e-3                      # drone_params.py:4
n_x, n_u = 6, 3                      # :6-7
S_DEFAULT, M_DEFAULT = 20, 50        # :9-10
T = 50.0                             # :11
DT_MODULE = T / S_DEFAULT            # :12  (the sampler's default dt, 2.5)
R = np.eye(n_u)                      # :13
FEEDBACK_GAIN = -np.hstack([0.05 * np.eye(3), 0.25 * np.eye(3)])  # :14-19
u_max = 10.0                         # :21
mass_nom, mass_delta = 32.0, 3.0     # :22-23
beta = 1e-2                          # :24
drag_coefficient = 0.2               # :25
obs_positions = np.array([[-1.4, -0.1, 0.0],
                          [-0.7, 0.3, 0.0],
                          [-0.3, 0.25, 0.0]])   # :34-37
obs_radii = np.array([0.3, 0.2, 0.2])           # :38-41
obs_radii_deltas = 0.025                        # :42
n_obs = 3
x_init = np.array([-1.9, 0.05, 0.2, 0.0, 0.0, 0.0])  # :45
x_final = np.zeros(n_x)


def sample_uncertain_parameters(rng, method='saa', M=100, S=S_DEFAULT,
                                dt=DT_MODULE):
    """drone_utils.py:61-93, replaying the reference's draw order on ``rng``
    (a ``np.random.RandomState``; ``RandomState(0)`` == ``np.random.seed(0)``).

    Order: uniform masses (M) -> for obs, for dim: uniform radius deltas (M)
    ['saa' only] -> randn in (M, S, 6) C order.  The nested per-(i,t)
    ``randn(6)`` calls of :88-90 are one bulk ``randn(M, S, 6)`` (same stream).
    Note ``dt`` defaults to the MODULE dt=2.5 (drone_utils.py:61), whatever S
    the Model later uses.
    """
    if method == 'saa':
        masses = rng.uniform(mass_nom - mass_delta, mass_nom + mass_delta, M)   # :64-66
        obs_Qs = np.zeros((M, n_obs, 3, 3))
        for obs_i in range(n_obs):                                               # :69-76
            for dim in range(3):
                delta_r = rng.uniform(-obs_radii_deltas, obs_radii_deltas, M)
                length = obs_radii[obs_i] + delta_r
                obs_Qs[:, obs_i, dim, dim] = 1.0 / length**2
    elif method == 'baseline':
        masses = rng.uniform(mass_nom, mass_nom, M)                              # :78-80 (consumes RNG)
        obs_Qs = np.zeros((M, n_obs, 3, 3))
        for obs_i in range(n_obs):                                               # :81-85
            for dim in range(3):
                obs_Qs[:, obs_i, dim, dim] = 1.0 / obs_radii[obs_i]**2
    else:
        raise ValueError(method)
    DWs = np.sqrt(dt) * rng.randn(M, S, n_x)                                     # :87-90
    if method == 'baseline':
        DWs = 0 * DWs                                                            # :91-92
    return DWs, masses, obs_Qs


class Model:
    """drone_risk.py:70-374 (the L1/L2/L3 layers; OSQP glue is not here)."""

    def __init__(self, S, DWs, masses, obs_Qs, method='saa', alpha=0.1):
        # drone_risk.py:71-93
        self.method, self.S, self.dt = method, S, T / S
        self.u_max, self.u_min = u_max, -u_max
        self.alpha, self.beta, self.drag_coefficient = alpha, beta, drag_coefficient
        self.DWs = np.asarray(DWs, dtype=np.float64)
        self.masses = np.asarray(masses, dtype=np.float64)
        self.obs_Qs = np.asarray(obs_Qs, dtype=np.float64)
        self.M = self.masses.shape[0]

    # -- layout helpers: drone_risk.py:95-106 --------------------------------
    def convert_us_vec_to_us_mat(self, us_vec):
        return np.reshape(us_vec, (n_u, self.S), 'F').T.copy()

    def convert_us_mat_to_us_vec(self, us_mat):
        return np.reshape(us_mat, (self.S * n_u), 'C')

    def initial_guess_us_mat(self):
        # drone_risk.py:108-120: only the first n_u-1 controls get 0.01
        us = np.zeros((self.S, n_u))
        us[:, :(n_u - 1)] = (self.u_max + self.u_min) / 2.0 + 1e-2
        return us

    # -- dynamics: drone_risk.py:122-137 ------------------------------------
    def b(self, x, u, mass):
        """x (...,6), u (...,3) or (3,), mass (...,) -> (...,6)."""
        v = x[..., 3:6]
        control_applied = u + x @ FEEDBACK_GAIN.T
        m = np.asarray(mass)[..., None]
        bvec = np.empty_like(x)
        bvec[..., :3] = v
        bvec[..., 3:6] = control_applied / m - self.drag_coefficient * np.abs(v) * v / m
        return bvec

    def sigma_dW(self, mass, dW):
        """sigma(x,u,mass) @ dW with sigma = (beta/m) I on block [3:6,3:6] (:133-137)."""
        out = np.zeros_like(dW)
        out[..., 3:6] = (self.beta / np.asarray(mass)[..., None]) * dW[..., 3:6]
        return out

    # -- rollout: drone_risk.py:139-162 --------------------------------------
    def us_to_state_trajectories(self, us_mat, masses=None, DWs=None):
        masses = self.masses if masses is None else masses
        DWs = self.DWs if DWs is None else DWs
        S, dt = self.S, self.dt
        M = masses.shape[0]
        xs = np.zeros((M, S + 1, n_x))
        xs[:, 0, :] = x_init
        for t in range(S):
            xt = xs[:, t, :]
            bt_dt = dt * self.b(xt, us_mat[t], masses)
            st_DWt = np.sqrt(dt) * self.sigma_dW(masses, DWs[:, t, :])   # :151 (sqrt(dt) AGAIN)
            xs[:, t + 1, :] = xt + bt_dt + st_DWt
        return xs

    def us_to_state_trajectory(self, us_mat, mass, dWs):
        return self.us_to_state_trajectories(
            us_mat, np.array([mass], dtype=np.float64), np.asarray(dWs)[None])[0]

    # -- constraints: drone_risk.py:164-213 ----------------------------------
    def final_constraints(self, xs):
        return xs[..., -1, :] - x_final

    def obstacle_avoidance_constraints(self, xs, obs_Q):
        """xs (M,S+1,6), obs_Q (M,n_obs,3,3) -> g (M,n_obs,S); also accepts one
        sample ((S+1,6), (n_obs,3,3)) -> (n_obs,S).  g = 1 - d^T Q[:2,:2] d,
        d = p_{t+1}[:2] - o_j[:2]  (:169-213)."""
        single = xs.ndim == 2
        if single:
            xs, obs_Q = xs[None], obs_Q[None]
        p = xs[:, 1:, :2]                                  # (M,S,2)
        d = p[:, None, :, :] - obs_positions[None, :, None, :2]   # (M,n_obs,S,2)
        Q = obs_Q[:, :, :2, :2]                            # (M,n_obs,2,2)
        Qd = np.einsum('mjab,mjtb->mjta', Q, d)
        g = 1.0 - np.einsum('mjta,mjta->mjt', d, Qd)
        return g[0] if single else g

    # -- linearization: drone_risk.py:239-280 --------------------------------
    def sensitivities(self, us_mat, xs):
        """Forward control-sensitivities of the Euler–Maruyama map.

        Returns Phi (M, S+1, 3 axes, S columns, 2) with
        Phi[m,t,a,s,:] = d(p_a, v_a)_t / d u_{s,a}; axes decouple and
        Phi[:, t, :, s] == 0 for t <= s.
        A_t = [[1, dt], [-0.05 dt/m, 1 - dt (0.25 + 2 c_d |v_t|)/m]],  B = [0, dt/m].
        """
        S, dt, M = self.S, self.dt, self.M
        m = self.masses[:, None, None]                     # (M,1,1)
        Phi = np.zeros((M, S + 1, 3, S, 2))
        for t in range(S):
            v = xs[:, t, 3:6][:, :, None]                  # (M,3,1)
            a21 = -0.05 * dt / m
            a22 = 1.0 - dt * (0.25 + 2.0 * self.drag_coefficient * np.abs(v)) / m
            P, V = Phi[:, t, :, :, 0], Phi[:, t, :, :, 1]
            Phi[:, t + 1, :, :, 0] = P + dt * V
            Phi[:, t + 1, :, :, 1] = a21 * P + a22 * V
            Phi[:, t + 1, :, t, 0] = 0.0
            Phi[:, t + 1, :, t, 1] = (dt / self.masses)[:, None]
        return Phi

    def get_all_constraints_coeffs(self, us_mat):
        """All samples at once; returns the reference's per-sample tuple
        (v_final_du (M,6,3S), val_final_lower (M,6), val_final_upper (M,6),
         g_obs_du (M,n_obs,S,3S), g_up (M,n_obs,S))."""
        S, M = self.S, self.M
        xs = self.us_to_state_trajectories(us_mat)
        v_final = self.final_constraints(xs)                        # (M,6)
        g_obs = self.obstacle_avoidance_constraints(xs, self.obs_Qs)  # (M,n_obs,S)
        Phi = self.sensitivities(us_mat, xs)

        v_final_du = np.zeros((M, n_x, n_u * S))
        for a in range(3):
            v_final_du[:, a, a::n_u] = Phi[:, S, a, :, 0]
            v_final_du[:, 3 + a, a::n_u] = Phi[:, S, a, :, 1]

        # dg/dp = -(Q + Q^T) d
        p = xs[:, 1:, :2]
        d = p[:, None, :, :] - obs_positions[None, :, None, :2]     # (M,n_obs,S,2)
        Q = self.obs_Qs[:, :, :2, :2]
        Qs = Q + np.swapaxes(Q, -1, -2)
        w = -np.einsum('mjab,mjtb->mjta', Qs, d)                    # (M,n_obs,S,2)
        g_obs_du = np.zeros((M, n_obs, S, n_u * S))
        for a in range(2):
            # row (j,t) uses p_{t+1}: Phi[:, t+1, a, s, 0]
            g_obs_du[:, :, :, a::n_u] = w[:, :, :, a, None] * Phi[:, 1:, a, :, 0][:, None, :, :]

        us_vec = self.convert_us_mat_to_us_vec(us_mat)
        val_final = -v_final + v_final_du @ us_vec                  # :271
        g_up = -g_obs + g_obs_du @ us_vec                           # :278
        return v_final_du, val_final, val_final.copy(), g_obs_du, g_up

    def sample_means(self, us_mat):
        """drone_risk.py:294-296: mean over samples of the final-constraint
        linearization."""
        fdu, flo, fup, _, _ = self.get_all_constraints_coeffs(us_mat)
        return fdu.mean(axis=0), flo.mean(axis=0), fup.mean(axis=0)

    # -- dense QP rows: drone_risk.py:282-374 (small M only: O(M^2) memory) --
    def get_all_constraints_coeffs_all(self, us_mat):
        S, M = self.S, self.M
        final_du, final_low, final_up, g_obs_du, g_obs_up = \
            self.get_all_constraints_coeffs(us_mat)
        final_du = final_du.mean(axis=0)
        final_low = final_low.mean(axis=0)
        final_up = final_up.mean(axis=0)
        final_dparams = np.concatenate(
            (final_du, np.zeros((final_du.shape[0], M + 2))), axis=-1)
        MULT = 0.01
        R_s = n_obs * S
        if self.method == 'baseline':                               # :303-325
            obs_low = -np.inf * np.ones(M * R_s)
            obs_up = np.inf * np.ones(M * R_s)
            obs_dparams = np.zeros((M * R_s, n_u * S + M + 2))
            for i in range(M):
                lo, hi = i * R_s, (i + 1) * R_s
                obs_dparams[lo:hi, :n_u * S] = np.reshape(MULT * g_obs_du[i], (R_s, n_u * S), 'C')
                obs_up[lo:hi] = MULT * g_obs_up[i].flatten() - 1e-3
        else:                                                       # :327-368
            obs_low = -np.inf * np.ones(1 + M + M * R_s + 1)
            obs_up = np.inf * np.ones(1 + M + M * R_s + 1)
            obs_dparams = np.zeros((1 + M + M * R_s + 1, n_u * S + M + 2))
            obs_dparams[0, -1] = M * self.alpha
            obs_dparams[0, n_u * S:-1] = 1.0      # NB: includes the slack column (:337)
            obs_up[0] = 0.0
            for i in range(M):
                idx_yi = n_u * S + i
                obs_dparams[1 + i, idx_yi] = -1.0
                obs_up[1 + i] = 0.0
                obs_dparams[1 + i, -2] = -1.0
                lo, hi = 1 + M + i * R_s, 1 + M + (i + 1) * R_s
                obs_dparams[lo:hi, :n_u * S] = np.reshape(MULT * g_obs_du[i], (R_s, n_u * S), 'C')
                obs_dparams[lo:hi, idx_yi] = -MULT
                obs_up[lo:hi] = MULT * g_obs_up[i].flatten()
                obs_dparams[lo:hi, -1] = -MULT
            obs_dparams[-1, -2] = -1.0
            obs_up[-1] = 0.0
        A = np.vstack([final_dparams, obs_dparams])
        low = np.hstack([final_low, obs_low])
        up = np.hstack([final_up, obs_up])
        return A, low, up

    # -- Monte-Carlo validation: drone_risk.py:649-695 -----------------------
    def monte_carlo_cost(self, us_mat):
        return self.dt_cost() * float(np.sum(np.diag(R)[None, :] * us_mat * us_mat))

    def dt_cost(self):
        # NB drone_risk.py:654 multiplies by the MODULE dt (=T/20), not self.dt
        return DT_MODULE

    def monte_carlo_no_collisions_constraint_verification(self, us_mat):
        """(:656-662) -> (B_satisfied (M,) bool, max_constraint (M,))."""
        xs = self.us_to_state_trajectories(us_mat)
        g = self.obstacle_avoidance_constraints(xs, self.obs_Qs)
        Z = g.reshape(self.M, -1).max(axis=1) - OSQP_TOL
        return Z <= 1e-6, Z
