"""Drone problem constants — same names and values as the reference's
``drone/drone_params.py:1-45`` (plain NumPy instead of jax.numpy)."""
import numpy as np

OSQP_POLISH = True
OSQP_TOL = 1e-3
n_x = 6   # (px, py, pz, vx, vy, vz)
n_u = 3   # (ux, uy, uz)
S = 20
M = 50
T = 50.0
dt = T / S
R = np.eye(n_u)
feedback_gain = -np.hstack([0.05 * np.eye(n_u), 0.25 * np.eye(n_u)])
u_max = 10
mass_nom = 32.0
mass_delta = 3
beta = 1e-2
drag_coefficient = 0.2
obs_positions = np.array([[-1.4, -0.1, 0.0], [-0.7, 0.3, 0.0], [-0.3, 0.25, 0.0]])
obs_radii = np.array([0.3, 0.2, 0.2])
obs_radii_deltas = 0.025
n_obs = obs_positions.shape[0]
x_init = np.array([-1.9, 0.05, 0.2, 0.0, 0.0, 0.0])
x_final = np.zeros(n_x)
