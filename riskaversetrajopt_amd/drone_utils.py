"""Drone samplers.  ``sample_uncertain_parameters`` keeps the reference's
signature and RNG draw order (``drone/drone_utils.py:61-93``) so that
``np.random.seed(0)`` reproduces identical sample draws, but is vectorised (the
reference's per-(i,t) ``randn(6)`` loop is one bulk draw from the same stream).
``sample_uncertain_parameters_device`` draws synthetic batches directly in HBM
in the kernels' SoA layout for throughput runs."""
import numpy as np

from . import drone_params as P


def sample_uncertain_parameters(method='saa', M=100, S=P.S, dt=P.dt, rng=None):
    """-> (DWs (M,S,6), masses (M,), obs_Qs (M,n_obs,3,3)); ``rng`` defaults to
    the global ``np.random`` stream like the reference."""
    rng = np.random if rng is None else rng
    if method == 'saa':
        masses = rng.uniform(P.mass_nom - P.mass_delta, P.mass_nom + P.mass_delta, M)
        obs_Qs = np.zeros((M, P.n_obs, 3, 3))
        for obs_i in range(P.n_obs):
            for dim in range(3):
                obs_delta_r = rng.uniform(-P.obs_radii_deltas, P.obs_radii_deltas, M)
                obs_Qs[:, obs_i, dim, dim] = 1.0 / (P.obs_radii[obs_i] + obs_delta_r) ** 2
    elif method == 'baseline':
        masses = rng.uniform(P.mass_nom - 0 * P.mass_delta, P.mass_nom + 0 * P.mass_delta, M)
        obs_Qs = np.zeros((M, P.n_obs, 3, 3))
        for obs_i in range(P.n_obs):
            obs_Qs[:, obs_i, [0, 1, 2], [0, 1, 2]] = 1.0 / P.obs_radii[obs_i] ** 2
    else:
        raise ValueError(f"unknown method {method!r}")
    DWs = np.sqrt(dt) * rng.randn(M, S, P.n_x)
    if method == 'baseline':
        DWs = 0 * DWs
    return DWs, masses, obs_Qs


def sample_uncertain_parameters_device(M, S, dt=P.dt, seed=0, device='cuda:0'):
    """Synthetic batch with the reference's distributions drawn on the device,
    already in kernel layout with row stride ld = M rounded up to a multiple of 4:
    dW [S][3][ld], mass [ld], Qsym [n_obs][3][ld] (fp32; the ld-M padding samples are
    ordinary draws that the kernels ignore)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ld = (M + 3) // 4 * 4
    dW = torch.randn((S, 3, ld), generator=g, device=device, dtype=torch.float32) * float(np.sqrt(dt))
    mass = P.mass_nom + P.mass_delta * (2 * torch.rand(ld, generator=g, device=device, dtype=torch.float32) - 1)
    r = torch.as_tensor(P.obs_radii, dtype=torch.float32, device=device)[:, None, None] + \
        P.obs_radii_deltas * (2 * torch.rand((P.n_obs, 3, ld), generator=g, device=device, dtype=torch.float32) - 1)
    q = 1.0 / (r * r)                      # diag entries (x, y, z) per obstacle
    Qsym = torch.stack([q[:, 0], torch.zeros_like(q[:, 0]), q[:, 1]], dim=1).contiguous()
    return dW, mass, Qsym
