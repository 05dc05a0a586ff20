"""Driving problem constants — same names and values as the reference's
``car/driving_params.py:1-42`` (plain NumPy instead of jax.numpy)."""
import numpy as np

OSQP_POLISH = True
OSQP_TOL = 3e-4
n_x = 8   # (px_e, py_e, v_e, phi_e, px_ped, py_ped, vx_ped, vy_ped)
n_u = 2   # (a, omega)
S = 20
M = 50
T = 10.0
dt = T / S
R = np.diag(np.array([1.0, 1.0 / 3.0]))
u_max = 100
omega_speed_nom = 0.1
omega_speed_del = 0.075
omega_repulsive_nom = 0.05
omega_repulsive_del = 0.045
ego_width = 2.695
ego_height = 1.663
ped_radius = 0.5
min_separation_distance = ped_radius + np.sqrt(ego_width**2 + ego_height**2)
speed_ped_des = 1.3
speed_ego_init = 4
position_ego_init = np.array([-20.0, 0.0])
position_ped_init = np.array([0.0, -6.0])
velocity_ego_init = np.array([speed_ego_init, 0.0])
velocity_ped_init = np.array([0.0, speed_ped_des])
position_ego_goal = np.array([20.0, 0.1])
velocity_ego_goal = np.array([4.1, 0.0])
state_init = np.concatenate((position_ego_init, velocity_ego_init, position_ped_init, velocity_ped_init), axis=-1)
variance_ped_initial_state = np.diag(np.array([1e-1, 1e-1, 1e-4, 1e-4]) ** 2)
