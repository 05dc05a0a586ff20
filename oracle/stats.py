"""Oracle (test infrastructure) — Monte-Carlo risk statistics, NumPy fp64.

Pinned by executing the reference's own text — see ``oracle/__init__.py``.
"""
import numpy as np


def monte_carlo_var(Z_samples, alpha):
    """drone_main_plot.py:640-652: empirical VaR by sorting —
    ``sort(Z)[M - floor(alpha*M) - 1]``."""
    Z = np.asarray(Z_samples, dtype=np.float64)
    M = len(Z)
    xth = int(np.floor(alpha * M))
    return np.sort(Z)[M - xth - 1]


def monte_carlo_avar(Z_samples, alpha):
    """drone_risk.py:663-695 / driving.py:639-671 / hopper.py:926-958.

    The reference minimises  t + (1/(M alpha)) sum_i y_i  s.t. y_i >= 0,
    y_i >= Z_i - t  with OSQP (a dense 2M x (M+1) LP) and then evaluates the
    closed form  t* + mean(max(Z - t*, 0))/alpha  (:694) at OSQP's t*.
    The objective is the Rockafellar–Uryasev function F(t); any t with
    #{Z_i > t} <= alpha*M <= #{Z_i >= t} minimises it, and the empirical VaR
    above is such a t.  F is evaluated here at that exact minimiser, so this
    value is <= the reference's (equal up to OSQP's tolerance).
    """
    Z = np.asarray(Z_samples, dtype=np.float64)
    M = len(Z)
    # index M - floor(alpha M) - 1, clamped at 0: when floor(alpha M) == M the
    # minimiser is min(Z) (monte_carlo_var's own index would wrap to max(Z) there)
    t_risk = np.sort(Z)[max(M - int(np.floor(alpha * M)) - 1, 0)]
    return t_risk + np.mean(np.maximum(Z - t_risk, 0.0)) / alpha


def rockafellar_uryasev(Z_samples, alpha, t):
    """F(t) = t + mean(max(Z - t, 0))/alpha — the closed form of :694 at any t."""
    Z = np.asarray(Z_samples, dtype=np.float64)
    return t + np.mean(np.maximum(Z - t, 0.0)) / alpha
