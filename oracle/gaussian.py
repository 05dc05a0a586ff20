"""Oracle (test infrastructure) — the mean / covariance recursion of the reference's Gaussian-
linearization baseline, ``/root/reference/drone/drone_gaussian.py:135-227`` (NumPy fp64).

Only used for BASELINE config C1 ("plumbing": M=100, S=30, CPU only): the recursion is checked
against the sample moments of SAA rollouts.  The baseline itself (IPOPT NLP with risk-allocation
variables, :238-535) has no sample axis and is out of scope.
Pinned by executing the reference's own text (tests/golden/ref_gaussian_S30.npz, tests/test_reference_pin.py).
"""
import numpy as np

from . import drone as od

MASS_VARIANCE = (2 * od.mass_delta) ** 2 / 12.0        # drone_gaussian.py:80 (uniform on +-mass_delta)


def mean_trajectory(us_mat, S):
    """drone_gaussian.py:161-174: nominal-mass Euler rollout without noise -> (S+1, 6)."""
    dt = od.T / S
    xs = np.zeros((S + 1, od.n_x))
    xs[0] = od.x_init
    for t in range(S):
        x, v = xs[t], xs[t, 3:6]
        acc = (us_mat[t] + od.FEEDBACK_GAIN @ x) / od.mass_nom - od.drag_coefficient * np.abs(v) * v / od.mass_nom
        xs[t + 1] = x + dt * np.concatenate([v, acc])
    return xs


def covariance_trajectory(us_mat, S, mass_variance=MASS_VARIANCE, outer_product=False):
    """drone_gaussian.py:176-227:  Sig+ = A Sig A^T + dt sigma sigma^T + var_m * (b_dm @ b_dm.T).

    Reference quirk, reproduced by default (found by executing the reference's text): ``b_dm = dt * jacfwd(b, mass)``
    is a 1-D array of shape (6,), so ``b_dm @ b_dm.T`` (:207) is the INNER product — a scalar — and
    ``Sig_next += Sigma_due_to_mass`` (:212) adds that scalar to ALL 36 entries of the covariance.
    ``outer_product=True`` gives the rank-one term b_dm b_dm^T the comment at :190-191 describes (used only for
    the sample-moment sanity check of config C1)."""
    dt, m = od.T / S, od.mass_nom
    xs = mean_trajectory(us_mat, S)
    Sig = np.zeros((S + 1, od.n_x, od.n_x))
    for t in range(S):
        x, v = xs[t], xs[t, 3:6]
        b_dx = np.zeros((6, 6))
        b_dx[:3, 3:] = np.eye(3)
        b_dx[3:, :3] = -0.05 * np.eye(3) / m
        b_dx[3:, 3:] = np.diag((-0.25 - 2.0 * od.drag_coefficient * np.abs(v)) / m)
        A = np.eye(6) + dt * b_dx                                            # :200-201
        sig = np.zeros((6, 6))
        sig[3:, 3:] = (od.beta / m) * np.eye(3)
        Sigma_w = dt * sig @ sig.T                                           # :203-204
        b_dm = np.zeros(6)
        b_dm[3:] = dt * (-(us_mat[t] + od.FEEDBACK_GAIN @ x) / m**2 + od.drag_coefficient * np.abs(v) * v / m**2)
        mass_term = np.outer(b_dm, b_dm) if outer_product else float(b_dm @ b_dm)          # :207 (see docstring)
        Sig[t + 1] = A @ Sig[t] @ A.T + Sigma_w + mass_variance * mass_term               # :206-213
    return Sig
