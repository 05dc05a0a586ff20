"""CPU: the oracle against vectors produced by EXECUTING THE REFERENCE'S OWN CODE.

``tests/golden/ref_*.npz`` hold inputs and outputs of the reference's ``class Model`` methods and Monte-Carlo
closures (``/root/reference/{drone/drone_risk.py, car/driving.py, hopper/hopper.py, drone/drone_main_plot.py}``),
extracted with ``ast`` and run unmodified by ``tests/golden/make_reference_golden.py`` in the build container
against ``tests/golden/jax_standin.py`` (a torch-fp64 stand-in for the absent jax: array type, vmap, jacfwd,
jacrev, hessian).  The samples were drawn by the reference's own sampler under its own seed.  These tests assert
that the NumPy oracle (and the product's host-side QP assembly) reproduce those vectors to ~1e-12, i.e. the
oracle is pinned by the reference's arithmetic text, with the caveat that autodiff came from torch.func.
The fixtures are data; the reference's text is read at generation time only and is not stored here.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import drone as od, driving as ocar, gaussian as og, hopper as oh, stats as ostats
from _oracle_qp import DroneOracleQP, DrivingOracleQP

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(rtol=1e-11, atol=1e-12)


def csc(f, prefix):
    shape = tuple(int(v) for v in f[prefix + "_shape"])
    return sp.csc_matrix((f[prefix + "_data"], f[prefix + "_indices"], f[prefix + "_indptr"]), shape=shape)


def assert_same_sparse(A, B, rtol=1e-11, atol=1e-13):
    """same pattern (exact zeros dropped on both sides) and same values"""
    A, B = sp.csc_matrix(A), sp.csc_matrix(B)
    A.sort_indices()
    B.sort_indices()
    A.eliminate_zeros()
    B.eliminate_zeros()
    assert A.shape == B.shape
    np.testing.assert_array_equal(A.indptr, B.indptr)
    np.testing.assert_array_equal(A.indices, B.indices)
    np.testing.assert_allclose(A.data, B.data, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ drone
@pytest.mark.parametrize("name", ["ref_drone_S20_M16", "ref_drone_S50_M8"])
def test_drone_oracle_matches_reference_execution(name):
    f = np.load(os.path.join(G, name + ".npz"))
    S, M, alpha = int(f["S"]), int(f["M"]), float(f["alpha"])
    # identical sample draws: the oracle's vectorised sampler == the reference's nested loops (drone_utils.py:61-93)
    DWs, masses, obs_Qs = od.sample_uncertain_parameters(np.random.RandomState(0), 'saa', M=M, S=S)
    assert np.array_equal(DWs, f["DWs"]) and np.array_equal(masses, f["masses"]) and np.array_equal(obs_Qs, f["obs_Qs"])
    assert float(f["sampler_dt"]) == od.DT_MODULE
    o = od.Model(S, DWs, masses, obs_Qs, 'saa', alpha)
    ob = od.Model(S, DWs, masses, obs_Qs, 'baseline', alpha)
    np.testing.assert_array_equal(o.initial_guess_us_mat(), f["init_us"])          # u_z stays 0 (:119)
    for kind in ("init", "graze"):
        us = f[f"{kind}_us"]
        xs = o.us_to_state_trajectories(us)
        np.testing.assert_allclose(xs, f[f"{kind}_xs"], **TOL)
        np.testing.assert_allclose(o.obstacle_avoidance_constraints(xs, obs_Qs), f[f"{kind}_g"], **TOL)
        np.testing.assert_allclose(o.final_constraints(xs), f[f"{kind}_final_value"], **TOL)
        fdu, flo, fup, gdu, gup = o.get_all_constraints_coeffs(us)
        np.testing.assert_allclose(fdu, f[f"{kind}_final_du"], **TOL)
        np.testing.assert_allclose(flo, f[f"{kind}_final_low"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(fup, f[f"{kind}_final_up"], rtol=1e-11, atol=1e-11)
        scale = np.abs(f[f"{kind}_g_obs_du"]).max()
        np.testing.assert_allclose(gdu, f[f"{kind}_g_obs_du"], rtol=1e-11, atol=1e-13 * scale)
        assert np.array_equal(gdu == 0.0, f[f"{kind}_g_obs_du"] == 0.0)             # structural zeros are exact zeros
        np.testing.assert_allclose(gup, f[f"{kind}_g_up"], rtol=1e-11, atol=1e-10)
        ok, Z = o.monte_carlo_no_collisions_constraint_verification(us)
        np.testing.assert_allclose(Z, f[f"{kind}_Z"], **TOL)
        np.testing.assert_array_equal(ok, f[f"{kind}_satisfied"])
        np.testing.assert_allclose(o.monte_carlo_cost(us), float(f[f"{kind}_cost"]), rtol=1e-13)
        np.testing.assert_allclose(ostats.monte_carlo_var(Z, 0.3), float(f[f"{kind}_var"]), rtol=1e-12)
        # dense QP rows (drone_risk.py:282-374), 'saa' and 'baseline'
        if f"{kind}_all_A_data" in f:
            A, low, up = o.get_all_constraints_coeffs_all(us)
            assert_same_sparse(A, csc(f, f"{kind}_all_A"))
            np.testing.assert_allclose(low, f[f"{kind}_all_low"], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(up, f[f"{kind}_all_up"], rtol=1e-11, atol=1e-11)
            A, low, up = ob.get_all_constraints_coeffs_all(us)
            assert_same_sparse(A, csc(f, f"{kind}_base_A"))
            np.testing.assert_allclose(up, f[f"{kind}_base_up"], rtol=1e-11, atol=1e-11)
        # the product's sparse assembler (assemble.py) against the reference's dense pack + csr/csc conversion
        for it in (0, 2):
            if f"{kind}_qp{it}_A_data" not in f:
                continue
            A, l, u = DroneOracleQP(o).get_constraints_coeffs(us, it)
            assert_same_sparse(A, csc(f, f"{kind}_qp{it}_A"), rtol=1e-10, atol=1e-18)
            np.testing.assert_allclose(l, f[f"{kind}_qp{it}_l"], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(u, f[f"{kind}_qp{it}_u"], rtol=1e-11, atol=1e-11)
    P, q = DroneOracleQP(o).get_objective_coeffs()
    assert_same_sparse(P, csc(f, "P"))
    np.testing.assert_array_equal(q, f["q"])
    np.testing.assert_array_equal(o.convert_us_mat_to_us_vec(f["graze_us"]), f["graze_us_vec"])
    np.testing.assert_array_equal(o.convert_us_vec_to_us_mat(f["graze_us_vec"]), f["graze_us_roundtrip"])


# ---------------------------------------------------------------- driving
@pytest.mark.parametrize("name", ["ref_driving_S20_M16", "ref_driving_S40_M8"])
def test_driving_oracle_matches_reference_execution(name):
    f = np.load(os.path.join(G, name + ".npz"))
    S, M, alpha = int(f["S"]), int(f["M"]), float(f["alpha"])
    x0, ws, wr, DWs = ocar.sample_uncertain_parameters(np.random.RandomState(0), M, 'saa', S)
    np.testing.assert_allclose(x0, f["states_init"], rtol=0, atol=1e-15)      # std = sqrt(diag) matmul vs elementwise
    assert np.array_equal(ws, f["omegas_speed"]) and np.array_equal(wr, f["omegas_repulsive"])
    assert np.array_equal(DWs, f["DWs"])
    xb, wsb, wrb, DWb = ocar.sample_uncertain_parameters(np.random.RandomState(0), M, 'baseline', S)
    assert np.array_equal(xb, f["base_states_init"]) and np.array_equal(DWb, f["base_DWs"])
    o = ocar.Model(f["states_init"], ws, wr, DWs, 'saa', alpha)
    ob = ocar.Model(xb, wsb, wrb, DWb, 'baseline', alpha)
    np.testing.assert_array_equal(o.initial_guess_us_mat(), f["init_us"])
    for kind in ("init", "swerve"):
        us = f[f"{kind}_us"]
        xs = o.us_to_state_trajectories(us)
        np.testing.assert_allclose(xs, f[f"{kind}_xs"], **TOL)
        np.testing.assert_allclose(-o.separation_distances_at_all_times(xs), f[f"{kind}_g"], **TOL)
        np.testing.assert_allclose(o.final_constraints(xs), f[f"{kind}_final_value"], **TOL)
        fdu, flo, fup, gdu, gup = o.get_all_constraints_coeffs(us)
        np.testing.assert_allclose(fdu, f[f"{kind}_final_du"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(flo, f[f"{kind}_final_low"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gdu, f[f"{kind}_g_obs_du"], rtol=1e-9, atol=1e-13)
        assert np.array_equal(gdu == 0.0, f[f"{kind}_g_obs_du"] == 0.0)
        np.testing.assert_allclose(gup, f[f"{kind}_g_up"], rtol=1e-10, atol=1e-10)
        ok, Z = o.monte_carlo_separation_constraints_verification(us)
        np.testing.assert_allclose(Z, f[f"{kind}_Z"], **TOL)
        np.testing.assert_array_equal(ok, f[f"{kind}_satisfied"])
        np.testing.assert_allclose(o.monte_carlo_cost(us), float(f[f"{kind}_cost"]), rtol=1e-13)
        if f"{kind}_all_A_data" in f:
            A, low, up = o.get_all_constraints_coeffs_all(us)
            assert_same_sparse(A, csc(f, f"{kind}_all_A"), rtol=1e-9)
            np.testing.assert_allclose(up, f[f"{kind}_all_up"], rtol=1e-10, atol=1e-10)
            A, low, up = ob.get_all_constraints_coeffs_all(us)
            assert_same_sparse(A, csc(f, f"{kind}_base_A"), rtol=1e-9)
            np.testing.assert_allclose(up, f[f"{kind}_base_up"], rtol=1e-10, atol=1e-10)
        for it in (0, 1):
            if f"{kind}_qp{it}_A_data" not in f:
                continue
            A, l, u = DrivingOracleQP(o).get_constraints_coeffs(us, it)
            assert_same_sparse(A, csc(f, f"{kind}_qp{it}_A"), rtol=1e-9)
            l_ref = f[f"{kind}_qp{it}_l"]
            nan = np.isnan(l_ref)
            if it < 1:
                # Reference quirk (driving.py:411-415): `ls[n_x:] *= 0` on lower bounds that are -inf gives NaN
                # (-inf * 0) for every row after the 8th.  Those rows of A are zeroed by the same statement, so any
                # lower bound <= 0 states the same (vacuous) constraint; the assembler emits 0 instead of NaN.
                assert nan[8:-2 * S].all() and not nan[:8].any() and not nan[-2 * S:].any()
                assert np.all(l[nan] == 0.0) and A.tocsr()[np.flatnonzero(nan)].nnz == 0
                l_ref = np.where(nan, 0.0, l_ref)
            else:
                assert not nan.any()
            np.testing.assert_allclose(l, l_ref, rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(u, f[f"{kind}_qp{it}_u"], rtol=1e-10, atol=1e-10)
    P, q = DrivingOracleQP(o).get_objective_coeffs()
    assert_same_sparse(P, csc(f, "P"))
    np.testing.assert_array_equal(q, f["q"])


# ----------------------------------------------------------------- hopper
@pytest.mark.parametrize("name", ["ref_hopper_S30_M30", "ref_hopper_S60_M24"])
def test_hopper_oracle_matches_reference_execution(name):
    f = np.load(os.path.join(G, name + ".npz"))
    S, M, alpha = int(f["S"]), int(f["M"]), float(f["alpha"])
    a, th, tau = oh.sample_friction_fields(np.random.RandomState(1), M)
    assert np.array_equal(a, f["intensities"]) and np.array_equal(th, f["thetas"]) and np.array_equal(tau, f["taus"])
    o = oh.Model(a, th, tau, 'saa', alpha, S=S)
    ob = oh.Model(a, th, tau, 'baseline', alpha, S=S)
    assert (o.time_jump, o.time_land) == (int(f["time_jump"]), int(f["time_land"]))
    Z = f["Z"]
    np.testing.assert_allclose(o.slip_risk_constraints(Z), f["gs"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ob.slip_risk_constraints(Z), f["gs_baseline"], rtol=1e-12, atol=1e-13)
    px, forces = o.contact_inputs(Z)
    np.testing.assert_allclose(px, f["px"], rtol=1e-14, atol=0)
    np.testing.assert_array_equal(forces, f["forces"])
    np.testing.assert_allclose(oh.friction_at_px(px, a, th, tau), f["mu"], rtol=1e-13)
    xs_mat, _ = o.convert_z_to_xs_us_mats(Z)
    np.testing.assert_allclose(o.end_effector_position(xs_mat), f["ee"], rtol=1e-14, atol=1e-16)
    ok, Zs = o.no_slip_constraints_verification(px, forces)
    np.testing.assert_allclose(Zs, f["Zs"], rtol=1e-12, atol=1e-14)
    np.testing.assert_array_equal(ok, f["satisfied"])

    # Jacobian of the slip rows wrt the NLP variables (reference: jacrev(g), hopper.py:569), rebuilt from the
    # oracle's per-(sample, contact) partials and the end-effector chain factors
    C = len(px)
    h, dfz, dpx = o.slip_partials(px, forces)
    Jee, Hee = o.contact_chain(Z)
    steps = o.contact_steps()
    nX, nU = (S + 1) * oh.n_x, S * oh.n_u
    J = np.zeros((1 + M + M * C + 1, o.num_vars))
    J[0, nX + nU:nX + nU + M] = 1.0
    J[0, -1] = M * alpha
    J[1 + np.arange(M), nX + nU + np.arange(M)] = -1.0
    for i in range(M):
        for c, t in enumerate(steps):
            r = 1 + M + i * C + c
            for k, col in enumerate((0, 2, 3)):
                J[r, t * oh.n_x + col] = dpx[i, c] * Jee[c, k]
            J[r, nX + t * oh.n_u + 2] = 1.0
            J[r, nX + t * oh.n_u + 3] = dfz[i, c]
            J[r, nX + nU + i] = -1.0
            J[r, -2] = -1.0
            J[r, -1] = -1.0
    Jref = csc(f, "J").toarray()
    np.testing.assert_allclose(J, Jref, rtol=1e-10, atol=1e-13)

    # lambda-weighted Hessian of the slip rows (reference: hessian(lambda . g), hopper.py:575-579)
    lam = f["lam"]
    D1, D2 = o.slip_hessian_sums(px, forces, lam)
    H = np.zeros((o.num_vars, o.num_vars))
    lam_dpx = np.sum(lam * dpx, axis=0)
    for c, t in enumerate(steps):
        xi = [t * oh.n_x + col for col in (0, 2, 3)]
        blk = D2[c] * np.outer(Jee[c], Jee[c]) + lam_dpx[c] * Hee[c]
        for a_, ia in enumerate(xi):
            for b_, ib in enumerate(xi):
                H[ia, ib] += blk[a_, b_]
            fzcol = nX + t * oh.n_u + 3
            H[ia, fzcol] += D1[c] * Jee[c, a_]
            H[fzcol, ia] += D1[c] * Jee[c, a_]
    Href = csc(f, "H").toarray()
    np.testing.assert_allclose(H, Href, rtol=1e-9, atol=1e-11 * np.abs(Href).max())

    # the same two matrices from the oracle's own assemblers (what the device facade's Model.slip_jacobian /
    # Model.slip_hessian are checked against): the reference's sparsity PATTERN, entry for entry, and its values
    for A, ref, rtol in ((o.slip_jacobian(Z), csc(f, "J"), 1e-10), (o.slip_hessian(Z, lam), csc(f, "H"), 1e-9)):
        ref = ref.tocsc()
        ref.sort_indices()
        assert A.shape == ref.shape
        np.testing.assert_array_equal(A.indptr, ref.indptr)
        np.testing.assert_array_equal(A.indices, ref.indices)
        np.testing.assert_allclose(A.data, ref.data, rtol=rtol, atol=1e-11 * np.abs(ref.data).max())
    # 'baseline' rows (hopper.py:339-348): rows i C + c, no y / t columns; against the saa matrix' own rows (zeroed fields)
    Jb = ob.slip_jacobian(Z)
    assert Jb.shape == (M * C, o.num_vars) and Jb[:, nX + nU:nX + nU + M].nnz == 0 and Jb[:, -1].nnz == 0
    np.testing.assert_array_equal(Jb[:, -2].toarray().ravel(), -np.ones(M * C))
    hb, dfzb, dpxb = ob.slip_partials(px, forces)
    np.testing.assert_allclose(Jb[np.arange(M * C), np.tile(nX + steps * oh.n_u + 3, M)].A1, dfzb.ravel(), rtol=1e-14)


# --------------------------------------------------- drone, Gaussian-linearization recursion (BASELINE config C1)
def test_gaussian_recursion_matches_reference_execution():
    """drone_gaussian.py:161-227 executed: mean trajectory and covariance recursion, including the reference's
    ``b_dm @ b_dm.T`` on a 1-D array (an inner product: the mass term is a SCALAR added to every entry)."""
    f = np.load(os.path.join(G, "ref_gaussian_S30.npz"))
    S, us = int(f["S"]), f["us"]
    assert float(f["mass_variance"]) == og.MASS_VARIANCE
    np.testing.assert_allclose(og.mean_trajectory(us, S), f["xs"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(og.covariance_trajectory(us, S), f["Sigmas"], rtol=1e-11, atol=1e-16)
    assert f["b_dmass"].shape == (6,)                                  # 1-D: hence the inner product
    rank_one = og.covariance_trajectory(us, S, outer_product=True)
    assert np.abs(rank_one - f["Sigmas"]).max() > 0.1 * np.abs(f["Sigmas"]).max()      # the two forms really differ


def test_fixtures_were_generated_from_this_reference():
    """Every ref_*.npz records the sha256 of the reference files it was generated from (make_reference_golden.py);
    where /root/reference exists (the build container) the files must still hash to that -- a changed reference means
    the fixtures have to be regenerated, not trusted.  (On the GPU box there is no reference: the stored digests are
    only checked for presence.)"""
    import glob
    import hashlib
    ref = os.environ.get("RATO_REFERENCE", "/root/reference")
    names = {"drone__drone_risk_py": "drone/drone_risk.py", "drone__drone_main_plot_py": "drone/drone_main_plot.py",
             "drone__drone_params_py": "drone/drone_params.py", "drone__drone_utils_py": "drone/drone_utils.py",
             "car__driving_py": "car/driving.py", "car__driving_params_py": "car/driving_params.py",
             "hopper__hopper_py": "hopper/hopper.py", "drone__drone_gaussian_py": "drone/drone_gaussian.py"}
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_*.npz")))
    assert len(files) >= 7
    for f in files:
        z = np.load(f)
        keys = [k for k in z.files if k.startswith("ref_sha256__")]
        assert keys, f
        for k in keys:
            assert z[k].dtype == np.uint8 and z[k].shape == (32,)
            path = os.path.join(ref, names[k[len("ref_sha256__"):]])
            if os.path.exists(path):
                assert hashlib.sha256(open(path, "rb").read()).digest() == z[k].tobytes(), (f, path)
