"""GPU parity against the REFERENCE'S OWN EXECUTION: the HIP path (through the C ABI and the Model facades) vs
``tests/golden/ref_*.npz`` — inputs and outputs recorded by running the reference's ``class Model`` text
(``make_reference_golden.py``; jax replaced by a torch-fp64 stand-in, see tests/test_reference_pin.py).
fp64 reference -> fp32 device tolerances of tests/_tol.py."""
import os

import numpy as np
import torch
import pytest
import scipy.sparse as sp

from tests import _tol as tol

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def csc(f, prefix):
    shape = tuple(int(v) for v in f[prefix + "_shape"])
    A = sp.csc_matrix((f[prefix + "_data"], f[prefix + "_indices"], f[prefix + "_indptr"]), shape=shape)
    A.sort_indices()
    return A


def assert_qp_close(A, l, u, A_ref, l_ref, u_ref, n_bounds):
    """same sparsity pattern as the reference's csc (exact zeros dropped); values to fp32 accuracy per row"""
    A = sp.csc_matrix(A)
    A.sort_indices()
    assert A.shape == A_ref.shape
    np.testing.assert_array_equal(A.indptr, A_ref.indptr)
    np.testing.assert_array_equal(A.indices, A_ref.indices)
    rowmax = np.asarray(abs(A_ref).max(axis=1).todense()).ravel()
    err = np.abs(A.data - A_ref.data)
    assert np.all(err <= tol.JAC_REL_ROWMAX * rowmax[A_ref.indices] + 1e-12), err.max()
    fin = np.isfinite(l_ref)
    assert np.array_equal(np.isneginf(l), np.isneginf(l_ref))
    np.testing.assert_allclose(l[fin], l_ref[fin], rtol=1e-4, atol=2e-4)
    assert np.array_equal(np.isposinf(u), np.isposinf(u_ref))
    fin = np.isfinite(u_ref)
    np.testing.assert_allclose(u[fin], u_ref[fin], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["ref_drone_S20_M16", "ref_drone_S50_M8"])
def test_drone_vs_reference_execution(name):
    from riskaversetrajopt_amd import drone_risk
    f = np.load(os.path.join(G, name + ".npz"))
    S, M, alpha = int(f["S"]), int(f["M"]), float(f["alpha"])
    d = drone_risk.Model(S, f["DWs"], f["masses"], f["obs_Qs"], 'saa', alpha)
    np.testing.assert_array_equal(d.initial_guess_us_mat(), f["init_us"])
    for kind in ("init", "graze"):
        us = f[f"{kind}_us"]
        xs = d.us_to_state_trajectories(us)
        np.testing.assert_allclose(xs, f[f"{kind}_xs"], rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)
        np.testing.assert_allclose(d.obstacle_avoidance_constraints(f[f"{kind}_xs"], f["obs_Qs"]), f[f"{kind}_g"],
                                   rtol=tol.G_RTOL, atol=tol.G_ATOL)
        gdu, gup = d.get_all_constraints_coeffs_batched(us)
        tol.assert_jac_close(gdu, f[f"{kind}_g_obs_du"], what="g_obs_du")
        assert np.array_equal(gdu == 0.0, f[f"{kind}_g_obs_du"] == 0.0)
        tol.assert_gup_close(gup, f[f"{kind}_g_up"], rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL, what=f"{name} {kind} g_up")
        fdu, flo, fup = d.sample_means(us)
        np.testing.assert_allclose(fdu, f[f"{kind}_final_du"].mean(0), rtol=tol.MEAN_RTOL, atol=tol.MEAN_ATOL)
        np.testing.assert_allclose(flo, f[f"{kind}_final_low"].mean(0), rtol=tol.MEAN_RTOL, atol=2e-5)
        # one sample through the reference's per-sample signature (drone_risk.py:239-280)
        i = M // 2
        v_du, v_lo, v_up, g_du, g_up1 = d.get_all_constraints_coeffs(us, f["masses"][i], f["DWs"][i], f["obs_Qs"][i])
        tol.assert_jac_close(v_du, f[f"{kind}_final_du"][i], what="v_final_du")
        np.testing.assert_allclose(v_lo, f[f"{kind}_final_low"][i], rtol=1e-5, atol=2e-5)
        tol.assert_jac_close(g_du, f[f"{kind}_g_obs_du"][i], what="g_obs_du[i]")
        ok, Z = d.monte_carlo_no_collisions_constraint_verification(us)
        np.testing.assert_allclose(Z, f[f"{kind}_Z"], rtol=tol.G_RTOL, atol=tol.G_ATOL)
        tol.assert_satisfied_close(ok, f[f"{kind}_Z"])
        np.testing.assert_allclose(d.monte_carlo_cost(us), float(f[f"{kind}_cost"]), rtol=1e-12)
        st = d.monte_carlo_statistics(us, alpha=0.3)
        assert abs(st["var"] - float(f[f"{kind}_var"])) < tol.RISK_ATOL * max(1.0, abs(float(f[f"{kind}_var"])))
        for it in (0, 2):
            if f"{kind}_qp{it}_A_data" not in f:
                continue
            A, l, u = d.get_constraints_coeffs(us, it)
            assert_qp_close(A, l, u, csc(f, f"{kind}_qp{it}_A"), f[f"{kind}_qp{it}_l"], f[f"{kind}_qp{it}_u"], 3 * S)
    # the Monte-Carlo report's form (drone_risk.py:697-725): both control sequences in ONE call, against the recorded Z / flags
    Zb, rec = d.eval_batch_device(np.stack([f["init_us"], f["graze_us"]]), alpha=0.3)
    for k, kind in enumerate(("init", "graze")):
        Zk = Zb[k].double().cpu().numpy()
        np.testing.assert_allclose(Zk, f[f"{kind}_Z"], rtol=tol.G_RTOL, atol=tol.G_ATOL)
        tol.assert_satisfied_close(Zk <= 1e-6, f[f"{kind}_Z"])
        assert abs(float(rec[k, 0]) - float(f[f"{kind}_var"])) < tol.RISK_ATOL * max(1.0, abs(float(f[f"{kind}_var"])))
    P, q = d.get_objective_coeffs()
    Pr = csc(f, "P")
    assert (sp.csc_matrix(P) != Pr).nnz == 0 and np.array_equal(q, f["q"])


@pytest.mark.parametrize("name", ["ref_driving_S20_M16", "ref_driving_S40_M8"])
def test_driving_vs_reference_execution(name):
    from riskaversetrajopt_amd import driving
    f = np.load(os.path.join(G, name + ".npz"))
    S, M, alpha = int(f["S"]), int(f["M"]), float(f["alpha"])
    d = driving.Model(M, 'saa', alpha, S=S,
                      samples=(f["states_init"], f["omegas_speed"], f["omegas_repulsive"], f["DWs"]))
    np.testing.assert_array_equal(d.initial_guess_us_mat(), f["init_us"])
    for kind in ("init", "swerve"):
        us = f[f"{kind}_us"]
        np.testing.assert_allclose(d.us_to_state_trajectories(us), f[f"{kind}_xs"],
                                   rtol=tol.STATE_RTOL, atol=tol.STATE_ATOL)
        # reference signature: distances from given trajectories (driving.py:232-236), one sample and batched
        np.testing.assert_allclose(d.separation_distances_at_all_times(f[f"{kind}_xs"]), -f[f"{kind}_g"],
                                   rtol=tol.G_RTOL, atol=tol.G_ATOL)
        np.testing.assert_allclose(d.separation_distances_at_all_times(f[f"{kind}_xs"][1]), -f[f"{kind}_g"][1],
                                   rtol=tol.G_RTOL, atol=tol.G_ATOL)
        gdu, gup = d.get_all_constraints_coeffs_batched(us)
        tol.assert_jac_close(gdu, f[f"{kind}_g_obs_du"], what="g_obs_du")
        assert np.array_equal(gdu == 0.0, f[f"{kind}_g_obs_du"] == 0.0)
        tol.assert_gup_close(gup, f[f"{kind}_g_up"], rtol=tol.GUP_RTOL, atol=tol.GUP_ATOL, what=f"{name} {kind} g_up")
        fdu, flo, _ = d.sample_means(us)
        tol.assert_jac_close(fdu, f[f"{kind}_final_du"].mean(0), what="final_du")
        np.testing.assert_allclose(flo, f[f"{kind}_final_low"].mean(0), rtol=1e-5, atol=5e-5)
        ok, Z = d.monte_carlo_separation_constraints_verification(us)
        np.testing.assert_allclose(Z, f[f"{kind}_Z"], rtol=tol.G_RTOL, atol=tol.G_ATOL)
        tol.assert_satisfied_close(ok, f[f"{kind}_Z"])
        np.testing.assert_allclose(d.monte_carlo_cost(us), float(f[f"{kind}_cost"]), rtol=1e-12)
        for it in (0, 1):
            if f"{kind}_qp{it}_A_data" not in f:
                continue
            A, l, u = d.get_constraints_coeffs(us, it)
            l_ref = f[f"{kind}_qp{it}_l"]
            l_ref = np.where(np.isnan(l_ref), 0.0, l_ref)      # -inf * 0 of driving.py:413 (see test_reference_pin.py)
            assert_qp_close(A, l, u, csc(f, f"{kind}_qp{it}_A"), l_ref, f[f"{kind}_qp{it}_u"], 2 * S)
    # the Monte-Carlo report's form (driving.py:675-740): both control sequences in ONE call, against the recorded Z / flags
    Zb, _ = d.eval_batch_device(np.stack([f["init_us"], f["swerve_us"]]))
    for k, kind in enumerate(("init", "swerve")):
        Zk = Zb[k].double().cpu().numpy()
        np.testing.assert_allclose(Zk, f[f"{kind}_Z"], rtol=tol.G_RTOL, atol=tol.G_ATOL)
        tol.assert_satisfied_close(Zk <= 1e-6, f[f"{kind}_Z"])


@pytest.mark.parametrize("name", ["ref_hopper_S30_M30", "ref_hopper_S60_M24"])
def test_hopper_vs_reference_execution(name):
    from riskaversetrajopt_amd import hopper
    f = np.load(os.path.join(G, name + ".npz"))
    S, M, alpha = int(f["S"]), int(f["M"]), float(f["alpha"])
    fields = (f["intensities"], f["thetas"], f["taus"])
    d = hopper.Model(M, 'saa', alpha, S=S, fields=fields)
    db = hopper.Model(M, 'baseline', alpha, S=S, fields=fields)
    Z = f["Z"]
    np.testing.assert_allclose(d.slip_risk_constraints(Z), f["gs"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(db.slip_risk_constraints(Z), f["gs_baseline"], rtol=0, atol=2e-5)
    px, forces = d.contact_inputs(Z)
    np.testing.assert_allclose(px, f["px"], rtol=1e-14)
    h, dfz, dpx = d.slip_partials(px, forces)
    np.testing.assert_allclose(-dfz, f["mu"], rtol=0, atol=2e-6)                 # dh/dfz = -mu_i(px_c)
    ok, Zs = d.no_slip_constraints_verification(px, forces)
    np.testing.assert_allclose(Zs, f["Zs"], rtol=0, atol=2e-5)
    # the reference's jac_g for the slip rows (hopper.py:569) -- the WHOLE recorded matrix, every sample and contact: the
    # facade assembles it in the reference's row / column order with the values written on the device
    # (rato_hopper_emit_jacobian_values); same sparsity pattern entry for entry, values to the fp32 tolerance
    C = len(px)
    Jref = sp.csc_matrix((f["J_data"], f["J_indices"], f["J_indptr"]), shape=tuple(f["J_shape"]))
    Jref.sort_indices()
    J = d.slip_jacobian(Z)
    assert J.shape == Jref.shape and J.nnz == Jref.nnz
    np.testing.assert_array_equal(J.indptr, Jref.indptr)
    np.testing.assert_array_equal(J.indices, Jref.indices)
    np.testing.assert_allclose(J.data, Jref.data, rtol=1e-4, atol=3e-5)
    exact = np.isin(Jref.data, (1.0, -1.0, M * alpha))                       # the structural constants: exact
    np.testing.assert_array_equal(J.data[exact], Jref.data[exact])
    # structural form on the device: every entry present, nnz as the ABI states it; a second call reuses the value buffer
    vals, indices, indptr, shape = d.slip_jacobian_device(Z)
    assert vals.numel() == 8 * C * M + 2 * M + 1 == indptr[-1] and shape == tuple(Jref.shape)
    vals2, *_ = d.slip_jacobian_device(Z, out=vals.clone())
    assert torch.equal(vals, vals2)
    # 'baseline' rows (:339-348): rows i C + c, zeroed fields (mu = mu_nom, no px dependence)
    Jb = db.slip_jacobian(Z)
    assert Jb.shape == (M * C, d.num_vars) and Jb.nnz == 3 * M * C     # fx, fz = -mu_nom, slack; dh/dpx = 0 dropped
    tj, tl = int(f["time_jump"]), int(f["time_land"])
    steps = np.concatenate([np.arange(0, tj), np.arange(tl, S)])
    nX = (S + 1) * 8
    np.testing.assert_allclose(Jb[np.arange(M * C), np.tile(nX + steps * 4 + 3, M)].A1, -0.10, rtol=0, atol=1e-7)
    # lambda-weighted Hessian (hopper.py:575-580): the whole recorded matrix, pattern and values
    Href = sp.csc_matrix((f["H_data"], f["H_indices"], f["H_indptr"]), shape=tuple(f["H_shape"]))
    Href.sort_indices()
    lam = f["lam"]
    H = d.slip_hessian(Z, lam)
    assert H.shape == Href.shape
    np.testing.assert_array_equal(H.indptr, Href.indptr)
    np.testing.assert_array_equal(H.indices, Href.indices)
    scale = np.abs(Href.data).max()
    np.testing.assert_allclose(H.data, Href.data, rtol=1e-4, atol=2e-5 * scale)
    # the two-sum entry point agrees with the three-sum one
    D1, D2 = d.slip_hessian_sums(px, forces, lam)
    D3 = d.slip_hessian_sums3(px, forces, lam)
    np.testing.assert_allclose(D3[:, 0], D1, rtol=1e-6, atol=1e-7 * np.abs(D1).max())
    np.testing.assert_allclose(D3[:, 1], D2, rtol=1e-6, atol=1e-7 * np.abs(D2).max())
