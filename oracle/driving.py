"""Oracle (test infrastructure) — driving (car + pedestrian) SAA inner loop,
NumPy fp64.

Restates, vectorised over the sample axis, the arithmetic of
``/root/reference/car/driving.py`` (+ ``driving_params.py``).
Pinned by executing the reference's own text — see ``oracle/__init__.py``.

The control-Jacobian (reference: ``jax.jacfwd``, driving.py:267-276) is the
analytic forward-sensitivity recursion X_{t+1} = J_t X_t + B_t with the full
8x8 one-step Jacobian J_t written out below; checked against independent
autodiff and finite differences in ``tests/test_oracle_driving.py``.
"""
import numpy as np

# ---- constants: driving_params.py:1-42 -------------------------------------
OSQP_TOL = 3e-4                       # :4
n_x, n_u = 8, 2                       # :6-9
S_DEFAULT, M_DEFAULT = 20, 50         # :11-12
T = 10.0                              # :13
R = np.diag([1.0, 1.0 / 3.0])         # :15
u_max = 100.0                         # :17
omega_speed_nom, omega_speed_del = 0.1, 0.075          # :18-19
omega_repulsive_nom, omega_repulsive_del = 0.05, 0.045  # :20-21
ego_width, ego_height, ped_radius = 2.695, 1.663, 0.5   # :23-25
min_separation_distance = ped_radius + np.sqrt(ego_width**2 + ego_height**2)  # :26-27
speed_ped_des = 1.3                   # :29
speed_ego_init = 4.0                  # :30
state_init = np.array([-20.0, 0.0, speed_ego_init, 0.0,
                       0.0, -6.0, 0.0, speed_ped_des])   # :31-40
state_ego_goal = np.array([20.0, 0.1, 4.1, 0.0])         # :35-36, driving.py:217-220
# driving.py:50-51: elementwise sqrt of diag([1e-1,1e-1,1e-4,1e-4])**2
std_ped_initial_state = np.array([1e-1, 1e-1, 1e-4, 1e-4])
BETA = 3e-2                           # driving.py:94


def sample_uncertain_parameters(rng, M, method='saa', S=S_DEFAULT):
    """driving.py:84-120 (the sampling part of ``Model.__init__``), replaying
    the reference draw order on ``rng`` (``np.random.RandomState``):
    uniform omega_speed (M) -> uniform omega_repulsive (M) -> ['saa' only]
    M x randn(4) -> randn in (M,S,8) C order.  ``DWs = sqrt(dt) * randn``.
    'baseline' zeroes the noise AND the gains (:117-120).
    """
    dt = T / S
    omegas_speed = rng.uniform(omega_speed_nom - omega_speed_del,
                               omega_speed_nom + omega_speed_del, M)           # :95-97
    omegas_repulsive = rng.uniform(omega_repulsive_nom - omega_repulsive_del,
                                   omega_repulsive_nom + omega_repulsive_del, M)  # :98-100
    states_init = np.repeat(state_init[None, :], M, axis=0)                     # :104
    if method == 'saa':                                                         # :105-109
        states_init[:, 4:] += std_ped_initial_state[None, :] * rng.randn(M, 4)
    DWs = np.sqrt(dt) * rng.randn(M, S, n_x)                                    # :112-116
    if method == 'baseline':
        DWs = 0 * DWs
        omegas_speed = 0 * omegas_speed
        omegas_repulsive = 0 * omegas_repulsive
    return states_init, omegas_speed, omegas_repulsive, DWs


class Model:
    """driving.py:83-373 (L1/L2/L3; OSQP glue is not here)."""

    def __init__(self, states_init, omegas_speed, omegas_repulsive, DWs,
                 method='saa', alpha=0.05):
        self.method, self.alpha, self.beta = method, alpha, BETA
        self.u_max, self.u_min = u_max, -u_max
        self.states_init = np.asarray(states_init, dtype=np.float64)
        self.omegas_speed = np.asarray(omegas_speed, dtype=np.float64)
        self.omegas_repulsive = np.asarray(omegas_repulsive, dtype=np.float64)
        self.DWs = np.asarray(DWs, dtype=np.float64)
        self.M, self.S = self.DWs.shape[0], self.DWs.shape[1]
        self.dt = T / self.S

    def convert_us_vec_to_us_mat(self, us_vec):                   # :122-126
        return np.reshape(us_vec, (n_u, self.S), 'F').T.copy()

    def convert_us_mat_to_us_vec(self, us_mat):                   # :128-130
        return np.reshape(us_mat, (self.S * n_u), 'C')

    def initial_guess_us_mat(self):                               # :132-143 (both controls)
        return np.zeros((self.S, n_u)) + (self.u_max + self.u_min) / 2.0 + 1e-2

    # -- dynamics: driving.py:145-184 ----------------------------------------
    def force_on_pedestrian(self, x, omega_speed, omega_repulsive):
        delta = x[..., 0:2] - x[..., 4:6]
        force = -omega_repulsive[..., None] * delta
        force = force / np.linalg.norm(delta, axis=-1, keepdims=True)
        delta_speed = speed_ped_des - x[..., 7]                   # x[7] = pedestrian v_y
        return force + (omega_speed * delta_speed)[..., None]     # scalar added to BOTH components

    def b(self, x, u, omega_speed, omega_repulsive):
        F = self.force_on_pedestrian(x, omega_speed, omega_repulsive)
        out = np.empty_like(x)
        out[..., 0] = x[..., 2] * np.cos(x[..., 3])
        out[..., 1] = x[..., 2] * np.sin(x[..., 3])
        out[..., 2] = u[..., 0]
        out[..., 3] = u[..., 1]
        out[..., 4] = x[..., 6]
        out[..., 5] = x[..., 7]
        out[..., 6:8] = F
        return out

    # -- rollout: driving.py:186-214 -----------------------------------------
    def us_to_state_trajectories(self, us_mat):
        S, dt, M = self.S, self.dt, self.M
        xs = np.zeros((M, S + 1, n_x))
        xs[:, 0, :] = self.states_init
        for t in range(S):
            xt = xs[:, t, :]
            bt_dt = dt * self.b(xt, np.broadcast_to(us_mat[t], (M, n_u)),
                                self.omegas_speed, self.omegas_repulsive)
            st_DWt = np.zeros((M, n_x))
            st_DWt[:, 6:] = np.sqrt(dt) * self.beta * self.DWs[:, t, 6:]   # :200 (sqrt(dt) AGAIN)
            xs[:, t + 1, :] = xt + bt_dt + st_DWt
        return xs

    # -- constraints: driving.py:216-236 -------------------------------------
    def final_constraints(self, xs):
        return xs[..., -1, :4] - state_ego_goal

    def separation_distances_at_all_times(self, xs):
        delta = xs[..., 1:, 0:2] - xs[..., 1:, 4:6]
        return np.linalg.norm(delta, axis=-1) - min_separation_distance

    # -- linearization: driving.py:260-298 -----------------------------------
    def step_jacobian(self, xt):
        """d x_{t+1} / d x_t of the Euler–Maruyama step, (M,8,8)."""
        dt, M = self.dt, self.M
        J = np.zeros((M, n_x, n_x))
        J[:, np.arange(n_x), np.arange(n_x)] = 1.0
        v, phi = xt[:, 2], xt[:, 3]
        J[:, 0, 2] = dt * np.cos(phi)
        J[:, 0, 3] = -dt * v * np.sin(phi)
        J[:, 1, 2] = dt * np.sin(phi)
        J[:, 1, 3] = dt * v * np.cos(phi)
        J[:, 4, 6] = dt
        J[:, 5, 7] = dt
        delta = xt[:, 0:2] - xt[:, 4:6]
        r = np.linalg.norm(delta, axis=-1)
        n = delta / r[:, None]
        H = (np.eye(2)[None] - n[:, :, None] * n[:, None, :]) / r[:, None, None]
        wr = self.omegas_repulsive[:, None, None]
        J[:, 6:8, 0:2] += -dt * wr * H          # dF/dp_ego
        J[:, 6:8, 4:6] += dt * wr * H           # dF/dp_ped
        J[:, 6, 7] += -dt * self.omegas_speed   # dF_k/dvy_ped (both k)
        J[:, 7, 7] += -dt * self.omegas_speed
        return J

    def sensitivities(self, xs):
        """X (M,S+1,8,2S): X[m,t,:,s*2+i] = d x_t / d u_{s,i}."""
        S, dt, M = self.S, self.dt, self.M
        X = np.zeros((M, S + 1, n_x, n_u * S))
        for t in range(S):
            J = self.step_jacobian(xs[:, t, :])
            X[:, t + 1] = J @ X[:, t]
            X[:, t + 1, 2, 2 * t] += dt
            X[:, t + 1, 3, 2 * t + 1] += dt
        return X

    def get_all_constraints_coeffs(self, us_mat):
        """All samples at once -> (v_final_du (M,4,2S), val_final_lower (M,4),
        val_final_upper (M,4), g_obs_du (M,S,2S), g_up (M,S))."""
        xs = self.us_to_state_trajectories(us_mat)
        v_final = self.final_constraints(xs)
        g_obs = -self.separation_distances_at_all_times(xs)       # :269
        X = self.sensitivities(xs)
        v_final_du = X[:, self.S, :4, :]
        delta = xs[:, 1:, 0:2] - xs[:, 1:, 4:6]
        n = delta / np.linalg.norm(delta, axis=-1, keepdims=True)  # (M,S,2)
        g_obs_du = -np.einsum('mta,mtac->mtc', n, X[:, 1:, 0:2, :] - X[:, 1:, 4:6, :])
        us_vec = self.convert_us_mat_to_us_vec(us_mat)
        val_final = -v_final + v_final_du @ us_vec                # :288
        g_up = -g_obs + g_obs_du @ us_vec                         # :295
        return v_final_du, val_final, val_final.copy(), g_obs_du, g_up

    def sample_means(self, us_mat):                               # :311-313
        fdu, flo, fup, _, _ = self.get_all_constraints_coeffs(us_mat)
        return fdu.mean(axis=0), flo.mean(axis=0), fup.mean(axis=0)

    # -- dense QP rows: driving.py:301-373 (small M only) --------------------
    def get_all_constraints_coeffs_all(self, us_mat):
        S, M = self.S, self.M
        final_du, final_low, final_up, gs_du, gs_up = self.get_all_constraints_coeffs(us_mat)
        final_du = final_du.mean(axis=0)
        final_low = final_low.mean(axis=0)
        final_up = final_up.mean(axis=0)
        final_dparams = np.concatenate((final_du, np.zeros((4, M + 2))), axis=-1)
        if self.method == 'baseline':                             # :320-329
            obs_low = -np.inf * np.ones(M * S)
            obs_up = np.inf * np.ones(M * S)
            obs_dparams = np.zeros((M * S, n_u * S + M + 2))
            for i in range(M):
                obs_dparams[i * S:(i + 1) * S, :n_u * S] = gs_du[i]
                obs_up[i * S:(i + 1) * S] = gs_up[i]
        else:                                                     # :331-368
            obs_low = -np.inf * np.ones(1 + M + M * S + 1)
            obs_up = np.inf * np.ones(1 + M + M * S + 1)
            obs_dparams = np.zeros((1 + M + M * S + 1, n_u * S + M + 2))
            obs_dparams[0, -1] = M * self.alpha
            obs_dparams[0, n_u * S:-1] = 1.0          # y columns AND the slack column
            obs_up[0] = 0.0
            for i in range(M):
                idx_yi = n_u * S + i
                obs_dparams[1 + i, idx_yi] = -1.0
                obs_up[1 + i] = 0.0
                obs_dparams[1 + i, -2] = -1.0
                lo, hi = 1 + M + i * S, 1 + M + (i + 1) * S
                obs_dparams[lo:hi, :n_u * S] = gs_du[i]
                obs_dparams[lo:hi, idx_yi] = -1.0
                obs_up[lo:hi] = gs_up[i]
                obs_dparams[lo:hi, -1] = -1.0
            obs_dparams[-1, -2] = -1.0
            obs_up[-1] = 0.0
        A = np.vstack([final_dparams, obs_dparams])
        low = np.hstack([final_low, obs_low])
        up = np.hstack([final_up, obs_up])
        return A, low, up

    # -- Monte-Carlo validation: driving.py:623-638 --------------------------
    def monte_carlo_cost(self, us_mat):
        # :623-629 multiplies by the module's dt -- the ONE dt of driving.py (driving_params.py:14, dt = T / S), the
        # same one the rollout uses; pinned by executing the reference at S = 40 (tests/test_reference_pin.py)
        return self.dt * float(np.sum(np.diag(R)[None, :] * us_mat * us_mat))

    def monte_carlo_separation_constraints_verification(self, us_mat):
        xs = self.us_to_state_trajectories(us_mat)
        val_obs = -self.separation_distances_at_all_times(xs)
        Z = val_obs.max(axis=1) - OSQP_TOL
        return Z <= 1e-6, Z
