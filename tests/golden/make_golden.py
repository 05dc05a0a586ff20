"""Generates the golden fixtures under tests/golden/ from the fp64 oracle.

Regression vectors of the oracle itself (``oracle/`` is pinned by executing the
reference's own text: make_reference_golden.py -> ref_*.npz, tests/test_reference_pin.py;
and checked against independent autodiff and finite differences by tests/test_oracle_*.py).  Inputs replay the
reference's RNG draw order (np.random.RandomState(seed)).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import drone as od          # noqa: E402
from oracle import driving as ocar      # noqa: E402
from oracle import hopper as oh         # noqa: E402
from oracle import stats as ostats      # noqa: E402


def drone_us(S, kind):
    if kind == 'init':
        us = np.zeros((S, 3))
        us[:, :2] = 0.01
        return us
    t = np.arange(S)[:, None]            # obstacle-grazing iterate
    return np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)


def car_us(S, kind):
    if kind == 'init':
        return np.zeros((S, 2)) + 0.01
    t = np.arange(S)[:, None]
    return np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01]) * (20.0 / S)


def make_drone():
    for S, M in ((20, 16), (50, 8)):
        rng = np.random.RandomState(0)
        DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
        model = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
        out = dict(S=S, M=M, DWs=DWs, masses=masses, obs_Qs=obs_Qs, alpha=0.1)
        for kind in ('init', 'graze'):
            us = drone_us(S, kind)
            xs = model.us_to_state_trajectories(us)
            fdu, flo, fup, gdu, gup = model.get_all_constraints_coeffs(us)
            ok, Z = model.monte_carlo_no_collisions_constraint_verification(us)
            out.update({f"{kind}_us": us, f"{kind}_xs": xs,
                        f"{kind}_g": model.obstacle_avoidance_constraints(xs, obs_Qs),
                        f"{kind}_final_du": fdu, f"{kind}_final_low": flo,
                        f"{kind}_g_obs_du": gdu, f"{kind}_g_up": gup,
                        f"{kind}_final_du_mean": fdu.mean(0), f"{kind}_final_low_mean": flo.mean(0),
                        f"{kind}_Z": Z, f"{kind}_satisfied": ok,
                        f"{kind}_var": ostats.monte_carlo_var(Z, 0.3), f"{kind}_avar": ostats.monte_carlo_avar(Z, 0.3)})
        np.savez_compressed(os.path.join(HERE, f"drone_S{S}_M{M}.npz"), **out)


def make_driving():
    for S, M in ((20, 16), (40, 8)):
        rng = np.random.RandomState(0)
        x0, ws, wr, DWs = ocar.sample_uncertain_parameters(rng, M, 'saa', S)
        model = ocar.Model(x0, ws, wr, DWs, 'saa', 0.05)
        out = dict(S=S, M=M, states_init=x0, omegas_speed=ws, omegas_repulsive=wr, DWs=DWs, alpha=0.05)
        for kind in ('init', 'swerve'):
            us = car_us(S, kind)
            xs = model.us_to_state_trajectories(us)
            fdu, flo, fup, gdu, gup = model.get_all_constraints_coeffs(us)
            ok, Z = model.monte_carlo_separation_constraints_verification(us)
            out.update({f"{kind}_us": us, f"{kind}_xs": xs,
                        f"{kind}_g": -model.separation_distances_at_all_times(xs),
                        f"{kind}_final_du": fdu, f"{kind}_final_low": flo,
                        f"{kind}_g_obs_du": gdu, f"{kind}_g_up": gup,
                        f"{kind}_Z": Z, f"{kind}_satisfied": ok,
                        f"{kind}_var": ostats.monte_carlo_var(Z, 0.3), f"{kind}_avar": ostats.monte_carlo_avar(Z, 0.3)})
        np.savez_compressed(os.path.join(HERE, f"driving_S{S}_M{M}.npz"), **out)


def hopper_Z(model, seed=5):
    rng = np.random.RandomState(seed)
    S, M = model.S, model.M
    Z = np.zeros(model.num_vars)
    xs = np.zeros((S + 1, 8))
    xs[:, 0] = np.linspace(0, 0.15, S + 1)
    xs[:, 1] = 1.0
    xs[:, 2] = 0.2 * np.sin(np.linspace(0, 3, S + 1))
    xs[:, 3] = 0.9 + 0.1 * np.cos(np.linspace(0, 2, S + 1))
    us = np.zeros((S, 4))
    us[:, 3] = 32.0 + rng.randn(S)
    us[:, 2] = 0.08 * us[:, 3] + 0.3 * rng.randn(S)
    Z[:(S + 1) * 8] = xs.reshape(-1)
    Z[(S + 1) * 8:(S + 1) * 8 + S * 4] = us.reshape(-1)
    Z[(S + 1) * 8 + S * 4:-2] = 0.1 * rng.rand(M)
    Z[-2], Z[-1] = 0.03, -0.4
    return Z


def make_hopper():
    for S, M in ((30, 30), (60, 24)):
        rng = np.random.RandomState(1)
        a, th, tau = oh.sample_friction_fields(rng, M)
        model = oh.Model(a, th, tau, 'saa', 0.2, S=S)
        Z = hopper_Z(model)
        px, forces = model.contact_inputs(Z)
        h, dfz, dpx = model.slip_partials(px, forces)
        lam = np.random.RandomState(2).rand(*h.shape)
        D1, D2 = model.slip_hessian_sums(px, forces, lam)
        ok, Zs = model.no_slip_constraints_verification(px, forces)
        base = oh.Model(a, th, tau, 'baseline', 0.2, S=S)
        np.savez_compressed(os.path.join(HERE, f"hopper_S{S}_M{M}.npz"), S=S, M=M, alpha=0.2,
                            intensities=a, thetas=th, taus=tau, Z=Z, px=px, forces=forces,
                            gs=model.slip_risk_constraints(Z), gs_baseline=base.slip_risk_constraints(Z),
                            h=h, dh_dfz=dfz, dh_dpx=dpx, lam=lam, D1=D1, D2=D2, Zs=Zs, satisfied=ok,
                            var=ostats.monte_carlo_var(Zs, 0.2), avar=ostats.monte_carlo_avar(Zs, 0.2))


if __name__ == "__main__":
    make_drone()
    make_driving()
    make_hopper()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
