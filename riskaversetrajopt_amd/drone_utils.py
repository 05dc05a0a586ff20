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


def sample_uncertain_parameters_device(M, S, dt=None, seed=0, device='cuda:0', want_dW=True):
    """Synthetic batch with the reference's distributions (drone_utils.py:61-93) drawn ON THE DEVICE by the library's
    Philox4x32-10 sampler (rato_drone_sample), already in kernel layout with row stride ld = M rounded up to a
    multiple of 4: dW [S][3][ld], mass [ld], Qsym [n_obs][3][ld] (fp32; the ld-M padding samples are ordinary draws
    that the kernels ignore).  ``dt``: the sampler's dt, default T/S — the dt of the Model the batch is for (the
    reference's own default is its module-level dt = T/S with ITS S, drone_utils.py:61).  ``want_dW=False``: no
    noise array at all — ``Model.from_device(..., noise_seed=seed)`` regenerates it inside the rollout kernel."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    dt = P.T / S if dt is None else dt
    ld = (M + 3) // 4 * 4
    dev = torch.device(device)
    dW = torch.empty((S, 3, ld), dtype=torch.float32, device=dev) if want_dW else None
    mass = torch.empty(ld, dtype=torch.float32, device=dev)
    Qsym = torch.empty((P.n_obs, 3, ld), dtype=torch.float32, device=dev)
    radii = (C.c_float * 3)(*[float(r) for r in P.obs_radii])
    with torch.cuda.device(dev):
        _lib.check(lib.rato_drone_sample(ld, ld, S, float(dt), int(seed), float(P.mass_nom), float(P.mass_delta), radii,
                                         float(P.obs_radii_deltas), _lib.ptr(dW), _lib.ptr(mass), _lib.ptr(Qsym),
                                         _lib.current_stream()), "rato_drone_sample")
    return dW, mass, Qsym
