"""Pins the hopper oracle (hopper.py:68-81, 300-367, 901-925): autodiff of the
slip value, layout of the slip-risk rows, Hessian sums."""
import numpy as np
import torch
from torch.func import grad, hessian, vmap

from oracle import hopper as oh
from tests import _torch_forward as tf


def make_model(M=7, S=30, seed=1, method='saa', alpha=0.2):
    rng = np.random.RandomState(seed)
    return oh.Model(*oh.sample_friction_fields(rng, M), method=method, alpha=alpha, S=S)


def synthetic_Z(model, seed=5):
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


def test_sampler_order():
    rng = np.random.RandomState(1)
    a, th, tau = oh.sample_friction_fields(rng, 30)
    np.random.seed(1)
    a_ref = 0.025 * (np.sqrt(2 / 30) * np.random.uniform(0, 1, (30, 30)))
    th_ref = np.random.uniform(0, np.pi, (30, 30))
    tau_ref = np.random.uniform(0, 2 * np.pi, (30, 30))
    assert np.array_equal(a, a_ref) and np.array_equal(th, th_ref) and np.array_equal(tau, tau_ref)


def test_contact_phase_mask():
    for S in (30, 60):
        m = make_model(S=S)
        c = m.contact_steps()
        assert len(c) == 2 * S // 3
        assert c[0] == 0 and c[S // 3 - 1] == S // 3 - 1 and c[S // 3] == 2 * S // 3 and c[-1] == S - 1


def test_slip_rows_layout_and_values():
    model = make_model()
    Z = synthetic_Z(model)
    gs = model.slip_risk_constraints(Z)
    M, C = model.M, 20
    assert gs.shape == (1 + M + M * C + 1,)
    _, _, ys, slack, t_risk = model.convert_z_to_variables(Z)
    assert gs[0] == M * model.alpha * t_risk + ys.sum()
    np.testing.assert_array_equal(gs[1:1 + M], -ys)
    assert gs[-1] == 0.0
    px, forces = model.contact_inputs(Z)
    i, c = 3, 13
    a, th, tau = (torch.tensor(v[i]) for v in (model.intensities, model.thetas, model.taus))
    h = tf.hopper_slip_value(torch.tensor(px[c]), torch.tensor(forces[c, 0]), torch.tensor(forces[c, 1]), a, th, tau)
    np.testing.assert_allclose(gs[1 + M + i * C + c], h.item() - t_risk - ys[i] - slack, rtol=1e-13)
    base = make_model(method='baseline')
    gb = base.slip_risk_constraints(Z)
    assert gb.shape == (M * C,)
    np.testing.assert_allclose(gb[i * C + c], forces[c, 0] - 0.1 * forces[c, 1] - slack, rtol=1e-13)


def test_partials_and_hessian_sums_match_autodiff():
    model = make_model(M=9)
    Z = synthetic_Z(model)
    px, forces = model.contact_inputs(Z)
    h, dfz, dpx = model.slip_partials(px, forces)
    f = tf.hopper_slip_value
    A, TH, TAU = (torch.tensor(v) for v in (model.intensities, model.thetas, model.taus))
    PX, FX, FZ = torch.tensor(px), torch.tensor(forces[:, 0]), torch.tensor(forces[:, 1])
    over_c = lambda fn: vmap(fn, in_dims=(0, 0, 0, None, None, None))
    over_i = lambda fn: vmap(fn, in_dims=(None, None, None, 0, 0, 0))
    val = over_i(over_c(f))(PX, FX, FZ, A, TH, TAU).numpy()
    gpx = over_i(over_c(grad(f, argnums=0)))(PX, FX, FZ, A, TH, TAU).numpy()
    gfz = over_i(over_c(grad(f, argnums=2)))(PX, FX, FZ, A, TH, TAU).numpy()
    np.testing.assert_allclose(h, val, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(dpx, gpx, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dfz, gfz, rtol=1e-13, atol=1e-13)
    Hpp = over_i(over_c(hessian(f, argnums=0)))(PX, FX, FZ, A, TH, TAU).numpy()
    Hpf = over_i(over_c(grad(grad(f, argnums=0), argnums=2)))(PX, FX, FZ, A, TH, TAU).numpy()
    lam = np.random.RandomState(2).rand(*h.shape)
    D1, D2 = model.slip_hessian_sums(px, forces, lam)
    np.testing.assert_allclose(D1, (lam * Hpf).sum(0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(D2, (lam * Hpp).sum(0), rtol=1e-12, atol=1e-13)


def test_monte_carlo_verification():
    model = make_model(M=200)
    Z = synthetic_Z(model)
    px, forces = model.contact_inputs(Z)
    ok, Zs = model.no_slip_constraints_verification(px, forces)
    h = model.no_slip_values(px, forces)
    np.testing.assert_array_equal(Zs, h.max(1))
    np.testing.assert_array_equal(ok, Zs <= 1e-6)


def test_end_effector_chain_factors_match_finite_differences():
    """dp/d(x0,x2,x3) and its Hessian for p = x0 + x3 sin x2 (hopper.py:166-171): the sample-independent chain that
    carries the kernel's dh/dpx, d2h/dpx2 to the NLP variables."""
    from oracle import hopper as oh
    rng = np.random.RandomState(0)
    o = oh.Model(*oh.sample_friction_fields(np.random.RandomState(1), 4))
    x = rng.randn(7, 8)
    J, H = o.end_effector_x_derivatives(x)
    eps = 1e-6
    for col, k in enumerate((0, 2, 3)):
        xp, xm = x.copy(), x.copy()
        xp[:, k] += eps
        xm[:, k] -= eps
        fd = (o.end_effector_position(xp)[:, 0] - o.end_effector_position(xm)[:, 0]) / (2 * eps)
        np.testing.assert_allclose(J[:, col], fd, atol=1e-8)
        Jp, _ = o.end_effector_x_derivatives(xp)
        Jm, _ = o.end_effector_x_derivatives(xm)
        np.testing.assert_allclose(H[:, :, col], (Jp - Jm) / (2 * eps), atol=1e-7)


def test_slip_rows_chain_to_state_variables():
    """d h_ic / d (x0, x2, x3) at a contact step = dh_dpx[i,c] * J[c]  (the formula documented in contact_chain),
    against finite differences of the slip rows through the full variable vector."""
    from oracle import hopper as oh
    S, M = 30, 3
    o = oh.Model(*oh.sample_friction_fields(np.random.RandomState(1), M), method='saa', alpha=0.1, S=S)
    rng = np.random.RandomState(2)
    Z = rng.randn(o.num_vars) * 0.3
    px, forces = o.contact_inputs(Z)
    _, _, dh_dpx = o.slip_partials(px, forces)
    J, _ = o.contact_chain(Z)
    steps = np.concatenate([np.arange(0, o.time_jump), np.arange(o.time_land, S)])
    eps = 1e-6
    for c in (0, 5, len(steps) - 1):
        t = steps[c]
        for col, k in enumerate((0, 2, 3)):
            idx = t * oh.n_x + k                         # xs_vec is the 'F' flatten of (n_x, S+1): entry (k, t)
            Zp, Zm = Z.copy(), Z.copy()
            Zp[idx] += eps
            Zm[idx] -= eps
            hp = o.no_slip_values(*o.contact_inputs(Zp))[:, c]
            hm = o.no_slip_values(*o.contact_inputs(Zm))[:, c]
            np.testing.assert_allclose((hp - hm) / (2 * eps), dh_dpx[:, c] * J[c, col], rtol=1e-5, atol=1e-8)
