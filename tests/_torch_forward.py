"""Independent fp64 forward functions in torch, differentiated with
``torch.func.jacfwd`` — the stand-in for the reference's ``jax.jacfwd`` /
``jacrev`` / ``hessian`` (jax is not installable here).  Used ONLY to pin the
oracle's hand-derived Jacobians; written functionally (no in-place writes) so
``vmap``/``jacfwd`` work.  Mirrors drone_risk.py:122-213, driving.py:145-236,
hopper.py:75-81,300-323 one sample at a time.
"""
import math
import torch

F64 = torch.float64     # every tensor below is created explicitly in fp64 (no global default-dtype change)


# ------------------------------ drone --------------------------------------
def drone_forward(us_mat, mass, dWs, obs_Q, S, dt):
    """-> (val_final (6,), val_obs (n_obs,S)) for ONE sample."""
    K = -torch.cat([0.05 * torch.eye(3, dtype=F64), 0.25 * torch.eye(3, dtype=F64)], dim=1)
    obs_positions = torch.tensor([[-1.4, -0.1, 0.0], [-0.7, 0.3, 0.0], [-0.3, 0.25, 0.0]], dtype=F64)
    x = torch.tensor([-1.9, 0.05, 0.2, 0.0, 0.0, 0.0], dtype=F64)
    beta, cd = 1e-2, 0.2
    ps = []
    for t in range(S):
        v = x[3:6]
        acc = (us_mat[t] + K @ x) / mass - cd * torch.abs(v) * v / mass
        bvec = torch.cat([v, acc])
        smat_dw = torch.cat([torch.zeros(3, dtype=F64), (beta / mass) * dWs[t, 3:6]])
        x = x + dt * bvec + math.sqrt(dt) * smat_dw
        ps.append(x[:2])
    P = torch.stack(ps)                                   # (S,2) = p_{1..S}
    d = P[None, :, :] - obs_positions[:, None, :2]        # (n_obs,S,2)
    Q = obs_Q[:, :2, :2]
    g = 1.0 - torch.einsum('jta,jab,jtb->jt', d, Q, d)
    return x - torch.zeros(6, dtype=F64), g


# ------------------------------ driving ------------------------------------
def driving_forward(us_mat, state_init, omega_speed, omega_repulsive, dWs, S, dt, d_min):
    """-> (val_final (4,), val_obs (S,)) for ONE sample (val_obs = -distance)."""
    goal = torch.tensor([20.0, 0.1, 4.1, 0.0], dtype=F64)
    x = state_init
    beta = 3e-2
    gs = []
    for t in range(S):
        delta = x[0:2] - x[4:6]
        force = -omega_repulsive * delta / torch.linalg.norm(delta)
        force = force + omega_speed * (1.3 - x[7])
        bvec = torch.stack([x[2] * torch.cos(x[3]), x[2] * torch.sin(x[3]),
                            us_mat[t, 0], us_mat[t, 1], x[6], x[7], force[0], force[1]])
        noise = torch.cat([torch.zeros(6, dtype=F64), beta * dWs[t, 6:8]])
        x = x + dt * bvec + math.sqrt(dt) * noise
        gs.append(-(torch.linalg.norm(x[0:2] - x[4:6]) - d_min))
    return x[:4] - goal, torch.stack(gs)


# ------------------------------ hopper -------------------------------------
def hopper_slip_value(px, fx, fz, a, th, tau):
    """h = fx - mu(px) fz for ONE (sample, contact)."""
    mu = 0.10 + torch.sum(a * torch.cos(th * px + tau))
    return fx - mu * fz
