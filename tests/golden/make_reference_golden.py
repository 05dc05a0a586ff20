"""Executes the REFERENCE'S OWN CODE for the hot path and records its inputs/outputs as golden vectors.

Build container only (needs /root/reference; never runs on the GPU box, never imported by the product):

    python tests/golden/make_reference_golden.py            # rewrites tests/golden/ref_*.npz

How.  The reference's scripts cannot be imported as they stand: each one runs its whole experiment at import
time and needs jax / osqp / ipyopt / LaTeX (none installed, no network).  This script therefore
  1. registers ``jax_standin`` (torch-fp64-backed ``jnp`` / ``vmap`` / ``jacfwd`` / ``jacrev`` / ``hessian``) as ``jax``;
  2. imports the reference's pure-constant modules for real (``drone_params``, ``drone_utils``, ``driving_params``);
  3. reads ``drone/drone_risk.py``, ``car/driving.py``, ``hopper/hopper.py``, ``drone/drone_main_plot.py`` with ``ast``
     AT RUN TIME, keeps the module-level constant assignments that precede ``class Model`` (with S / M / dt
     overridden to the fixture's sizes — the only edit), the ``class Model`` definition, module-level ``def``s and the
     named Monte-Carlo closures (which the reference nests under ``if B_validate_monte_carlo:``), and ``exec``s that
     text unmodified.  Nothing of the reference's text is stored in this repository.
  4. calls the reference's methods on the fixture inputs (drawn by the reference's own sampler under its own seed)
     and writes inputs + outputs to ``tests/golden/ref_*.npz``.
``tests/test_reference_pin.py`` then asserts that the oracle reproduces these vectors to 1e-12 (CPU) and the GPU
tests compare the HIP path with them.  jax itself is a stand-in here: see jax_standin.py's docstring.
"""
import ast
import importlib
import os
import sys
import time
import warnings
from functools import partial
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("RATO_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)

import jax_standin  # noqa: E402

warnings.filterwarnings("ignore", category=DeprecationWarning)


# ---------------------------------------------------------------- extraction
class _Override(ast.NodeTransformer):
    def __init__(self, overrides):
        self.ov = overrides

    def visit_Assign(self, node):
        if len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id in self.ov:
            node.value = ast.Constant(self.ov[node.targets[0].id])
        return node


REF_SHA256 = {}           # reference file (relative path) -> sha256 of the bytes that were executed


def _assert_standin_covers(nodes, path):
    """The stand-in's ``vmap`` maps the LEADING axis of every positional argument and nothing else: refuse reference
    text that calls it any other way (in_axes / out_axes / axis_name / several functions), instead of silently
    computing something else."""
    for top in nodes:
        for node in ast.walk(top):
            if not isinstance(node, ast.Call):
                continue
            f = node.func
            name = f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else None)
            if name == "vmap":
                kws = {k.arg for k in node.keywords}
                assert not kws and len(node.args) == 1, (path, node.lineno, "vmap called with", kws or node.args)
            if name in ("jacfwd", "jacrev", "hessian", "grad"):
                kws = {k.arg for k in node.keywords}
                assert kws <= {"argnums"} and len(node.args) <= 2, (path, node.lineno, name, kws)


def load_reference(path, ns, overrides=(), nested=()):
    """exec the reference's constants + ``class Model`` + top-level defs from ``path`` into ``ns``; returns a
    callable that execs the named nested closures (found anywhere in the file) once ``ns`` holds their globals."""
    import hashlib
    raw = Path(path).read_bytes()
    REF_SHA256[os.path.relpath(path, REF)] = hashlib.sha256(raw).digest()
    src = raw.decode()
    tree = ast.parse(src, filename=path)
    keep, seen_model = [], False
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "Model":
            keep.append(node)
            seen_model = True
        elif isinstance(node, ast.FunctionDef):
            keep.append(node)
        elif isinstance(node, ast.Assign) and not seen_model and all(isinstance(t, ast.Name) for t in node.targets):
            keep.append(_Override(dict(overrides)).visit(node))
    assert seen_model, path
    _assert_standin_covers(keep, path)
    mod = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)

    top = {id(n) for n in tree.body}
    closures = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name in nested and id(n) not in top]
    found = {n.name for n in closures}
    assert found == set(nested), (path, set(nested) - found)
    _assert_standin_covers(closures, path)

    def define_closures():
        seen = set()
        body = [n for n in closures if not (n.name in seen or seen.add(n.name))]   # first definition of each name
        m = ast.Module(body=body, type_ignores=[])
        ast.fix_missing_locations(m)
        exec(compile(m, path, "exec"), ns)
    return define_closures


def base_namespace(jax):
    return {"np": np, "sp": sp, "jnp": jax.numpy, "jit": jax.jit, "vmap": jax.vmap, "jacfwd": jax.jacfwd,
            "jacrev": jax.jacrev, "hessian": jax.hessian, "grad": jax.grad, "partial": partial,
            "warn": warnings.warn, "time": time.time, "Path": Path, "print": lambda *a, **k: None}


def with_hashes(out, *rel_paths):
    """+ sha256 (uint8[32]) of every reference file this fixture was generated from, and of the modules imported for
    real: tests/test_reference_pin.py re-hashes /root/reference where it exists, so a changed reference is noticed."""
    import hashlib
    for rel in rel_paths:
        digest = REF_SHA256.get(rel) or hashlib.sha256(Path(os.path.join(REF, rel)).read_bytes()).digest()
        out["ref_sha256__" + rel.replace("/", "__").replace(".", "_")] = np.frombuffer(digest, dtype=np.uint8).copy()
    return out


def npy(x):
    import torch
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def csc_triplet(A, prefix):
    A = sp.csc_matrix(A)
    A.sort_indices()
    return {prefix + "_data": A.data, prefix + "_indices": A.indices, prefix + "_indptr": A.indptr,
            prefix + "_shape": np.array(A.shape)}


# ---------------------------------------------------------------- inputs (same iterates as make_golden.py)
def drone_us(S, kind):
    if kind == 'init':
        return None                                    # the reference's own initial_guess_us_mat()
    t = np.arange(S)[:, None]
    return np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)


def car_us(S, kind):
    if kind == 'init':
        return None
    t = np.arange(S)[:, None]
    return np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01]) * (20.0 / S)


# ---------------------------------------------------------------- drone
def make_drone(jax, S, M, alpha=0.1):
    jnp, vmap = jax.numpy, jax.vmap
    sys.path.insert(0, os.path.join(REF, "drone"))
    drone_params = importlib.import_module("drone_params")
    drone_utils = importlib.import_module("drone_utils")
    ns = base_namespace(jax)
    ns.update(drone_params=drone_params, sample_uncertain_parameters=drone_utils.sample_uncertain_parameters)
    closures = load_reference(os.path.join(REF, "drone", "drone_risk.py"), ns, overrides={"S": S, "M": M},
                              nested=("monte_carlo_cost", "monte_carlo_no_collisions_constraint_verification"))
    ns_plot = base_namespace(jax)
    ns_plot.update(drone_params=drone_params)
    var_closure = load_reference(os.path.join(REF, "drone", "drone_main_plot.py"), ns_plot, nested=("monte_carlo_var",))
    var_closure()

    np.random.seed(0)                                              # drone_risk.py:57
    DWs, masses, obs_Qs = ns["sample_uncertain_parameters"]('saa', M=M, S=S)      # drone_utils.py:61 (dt = module dt)
    Model = ns["Model"]
    model = Model(S, jnp.array(DWs), jnp.array(masses), jnp.array(obs_Qs), 'saa', alpha)
    base = Model(S, jnp.array(DWs), jnp.array(masses), jnp.array(obs_Qs), 'baseline', alpha)
    ns["model"] = model
    closures()
    full = S == 20
    out = dict(S=S, M=M, alpha=alpha, DWs=DWs, masses=masses, obs_Qs=obs_Qs, sampler_dt=drone_utils.dt)
    P, q = model.get_objective_coeffs()
    out.update(csc_triplet(P, "P"))
    out["q"] = q
    for kind in ("init", "graze"):
        us = drone_us(S, kind)
        us = model.initial_guess_us_mat() if us is None else jnp.array(us)
        xs = model.us_to_state_trajectories(us)
        g = vmap(model.obstacle_avoidance_constraints)(xs, model.obs_Qs)
        Us = jnp.repeat(us[None], M, axis=0)
        fdu, flo, fup, gdu, gup = vmap(model.get_all_constraints_coeffs)(Us, model.masses, model.DWs, model.obs_Qs)
        ok, Z = vmap(ns["monte_carlo_no_collisions_constraint_verification"])(Us, model.masses, model.DWs, model.obs_Qs)
        out.update({f"{kind}_us": npy(us), f"{kind}_xs": npy(xs), f"{kind}_g": npy(g),
                    f"{kind}_final_du": npy(fdu), f"{kind}_final_low": npy(flo), f"{kind}_final_up": npy(fup),
                    f"{kind}_g_obs_du": npy(gdu), f"{kind}_g_up": npy(gup),
                    f"{kind}_Z": npy(Z), f"{kind}_satisfied": npy(ok).astype(bool),
                    f"{kind}_cost": float(ns["monte_carlo_cost"](us)),
                    f"{kind}_var": float(ns_plot["monte_carlo_var"](npy(Z), 0.3)),
                    f"{kind}_final_value": npy(vmap(model.final_constraints)(xs))})
        if full or kind == "graze":                    # the matrices repeat the Jacobian: keep the big case small
            for it in ((0, 2) if full else (2,)):
                A, l, u = model.get_constraints_coeffs(us, it)
                out.update(csc_triplet(A, f"{kind}_qp{it}_A"))
                out.update({f"{kind}_qp{it}_l": l, f"{kind}_qp{it}_u": u})
        if full:
            A, low, up = model.get_all_constraints_coeffs_all(us)
            out.update(csc_triplet(npy(A), f"{kind}_all_A"))
            out.update({f"{kind}_all_low": npy(low), f"{kind}_all_up": npy(up)})
            A, low, up = base.get_all_constraints_coeffs_all(us)
            out.update(csc_triplet(npy(A), f"{kind}_base_A"))
            out.update({f"{kind}_base_low": npy(low), f"{kind}_base_up": npy(up)})
    vec = model.convert_us_mat_to_us_jaxvec(jnp.array(out["graze_us"]))
    out["graze_us_vec"] = npy(vec)
    out["graze_us_roundtrip"] = npy(model.convert_us_vec_to_us_mat(vec))
    out["L2_error"] = float(ns["L2_error_us"](out["graze_us"], out["init_us"]))
    np.savez_compressed(os.path.join(HERE, f"ref_drone_S{S}_M{M}.npz"),
                        **with_hashes(out, "drone/drone_risk.py", "drone/drone_main_plot.py", "drone/drone_params.py",
                                      "drone/drone_utils.py"))
    sys.path.pop(0)


# ---------------------------------------------------------------- driving
def make_driving(jax, S, M, alpha=0.05):
    jnp, vmap = jax.numpy, jax.vmap
    sys.path.insert(0, os.path.join(REF, "car"))
    driving_params = importlib.import_module("driving_params")
    ns = base_namespace(jax)
    ns.update(driving_params=driving_params)
    dt = driving_params.T / S                              # the reference computes dt = T / S (driving_params.py:14)
    closures = load_reference(os.path.join(REF, "car", "driving.py"), ns, overrides={"S": S, "M": M, "dt": dt},
                              nested=("monte_carlo_cost", "monte_carlo_separation_constraints_verification"))
    Model = ns["Model"]
    np.random.seed(0)                                      # driving.py:61
    model = Model(M, 'saa', alpha)
    np.random.seed(0)
    base = Model(M, 'baseline', alpha)
    ns["model"] = model
    closures()
    full = S == 20
    out = dict(S=S, M=M, alpha=alpha, dt=dt, states_init=npy(model.states_init), omegas_speed=npy(model.omegas_speed),
               omegas_repulsive=npy(model.omegas_repulsive), DWs=npy(model.DWs),
               base_states_init=npy(base.states_init), base_DWs=npy(base.DWs))
    P, q = model.get_objective_coeffs()
    out.update(csc_triplet(P, "P"))
    out["q"] = q
    for kind in ("init", "swerve"):
        us = car_us(S, kind)
        us = model.initial_guess_us_mat() if us is None else jnp.array(us)
        xs = model.us_to_state_trajectories(us)
        dist = vmap(model.separation_distances_at_all_times)(xs)
        Us = jnp.repeat(us[None], M, axis=0)
        fdu, flo, fup, gdu, gup = vmap(model.get_all_constraints_coeffs)(
            Us, model.states_init, model.omegas_speed, model.omegas_repulsive, model.DWs)
        ok, Z = vmap(ns["monte_carlo_separation_constraints_verification"])(
            Us, model.states_init, model.omegas_speed, model.omegas_repulsive, model.DWs)
        out.update({f"{kind}_us": npy(us), f"{kind}_xs": npy(xs), f"{kind}_g": -npy(dist),
                    f"{kind}_final_du": npy(fdu), f"{kind}_final_low": npy(flo), f"{kind}_final_up": npy(fup),
                    f"{kind}_g_obs_du": npy(gdu), f"{kind}_g_up": npy(gup),
                    f"{kind}_Z": npy(Z), f"{kind}_satisfied": npy(ok).astype(bool),
                    f"{kind}_cost": float(ns["monte_carlo_cost"](us)),
                    f"{kind}_final_value": npy(vmap(model.final_constraints)(xs))})
        if full or kind == "swerve":
            for it in ((0, 1) if full else (1,)):
                A, l, u = model.get_constraints_coeffs(us, it)
                out.update(csc_triplet(A, f"{kind}_qp{it}_A"))
                out.update({f"{kind}_qp{it}_l": l, f"{kind}_qp{it}_u": u})
        if full:
            A, low, up = model.get_all_constraints_coeffs_all(us)
            out.update(csc_triplet(npy(A), f"{kind}_all_A"))
            out.update({f"{kind}_all_low": npy(low), f"{kind}_all_up": npy(up)})
            A, low, up = base.get_all_constraints_coeffs_all(us)
            out.update(csc_triplet(npy(A), f"{kind}_base_A"))
            out.update({f"{kind}_base_low": npy(low), f"{kind}_base_up": npy(up)})
    np.savez_compressed(os.path.join(HERE, f"ref_driving_S{S}_M{M}.npz"),
                        **with_hashes(out, "car/driving.py", "car/driving_params.py"))
    sys.path.pop(0)


# ---------------------------------------------------------------- hopper
def hopper_Z(S, M, num_vars, seed=5):
    """A synthetic NLP iterate (same construction as make_golden.py)."""
    rng = np.random.RandomState(seed)
    Z = np.zeros(num_vars)
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


def make_hopper(jax, S, M, alpha=0.2):
    jnp, vmap = jax.numpy, jax.vmap
    ns = base_namespace(jax)
    np.random.seed(1)                                      # hopper.py:33; the fields are drawn at module level (:70-74)
    closures = load_reference(os.path.join(REF, "hopper", "hopper.py"), ns,
                              overrides={"S": S, "M": M, "time_jump": S // 3, "time_land": 2 * S // 3},
                              nested=("no_slip_constraint", "no_slip_constraints_verification"))
    closures()
    Model = ns["Model"]
    model, base = Model(M, 'saa', alpha), Model(M, 'baseline', alpha)
    Z = hopper_Z(S, M, ns["num_vars"])
    Zt = jnp.array(Z)
    gs = model.slip_risk_constraints(Zt)
    gs_base = base.slip_risk_constraints(Zt)
    J = jax.jacrev(model.slip_risk_constraints)(Zt)                    # hopper.py:569 restricted to the slip rows
    C = (len(npy(gs)) - 2 - M) // M
    lam = np.random.RandomState(2).rand(M, C)
    lam_full = np.zeros(len(npy(gs)))
    lam_full[1 + M:1 + M + M * C] = lam.reshape(-1)
    lam_t = jnp.array(lam_full)
    H = jax.hessian(lambda z: jnp.dot(lam_t, model.slip_risk_constraints(z)))(Zt)     # hopper.py:575-579
    xs_mat, us_mat = model.convert_z_to_xs_us_mats(Zt)
    tj, tl = ns["time_jump"], ns["time_land"]
    ee_x = vmap(model.end_effector_position)(xs_mat)[:, 0]
    px = jnp.concatenate([ee_x[:tj], ee_x[tl:-1]], axis=0)                            # as gathered at :305-311
    forces = jnp.concatenate([us_mat[:tj, 2:], us_mat[tl:, 2:]], axis=0)
    intens, th, tau = jnp.array(ns["intensities"]), jnp.array(ns["thetas"]), jnp.array(ns["taus"])
    ok, Zs = vmap(ns["no_slip_constraints_verification"])(
        jnp.repeat(px[None], M, axis=0), jnp.repeat(forces[None], M, axis=0), intens, th, tau)
    mu = vmap(lambda a, b, c: vmap(lambda p: ns["friction_at_px"](p, a, b, c))(px))(intens, th, tau)
    Jn, Hn = npy(J), npy(H)
    out = dict(S=S, M=M, alpha=alpha, intensities=ns["intensities"], thetas=ns["thetas"], taus=ns["taus"], Z=Z,
               px=npy(px), forces=npy(forces), gs=npy(gs), gs_baseline=npy(gs_base), lam=lam, mu=npy(mu),
               Zs=npy(Zs), satisfied=npy(ok).astype(bool), time_jump=tj, time_land=tl,
               ee=npy(vmap(model.end_effector_position)(xs_mat)), initial_guess=model.initial_guess())
    out.update(csc_triplet(Jn, "J"))
    out.update(csc_triplet(Hn, "H"))
    np.savez_compressed(os.path.join(HERE, f"ref_hopper_S{S}_M{M}.npz"), **with_hashes(out, "hopper/hopper.py"))


# ---------------------------------------------------------------- drone, Gaussian-linearization baseline (config C1)
def make_gaussian(jax, S, M_unused, alpha=0.1):
    """drone_gaussian.py:161-227: the mean trajectory and the covariance recursion (no sample axis)."""
    jnp = jax.numpy
    sys.path.insert(0, os.path.join(REF, "drone"))
    drone_params = importlib.import_module("drone_params")
    drone_utils = importlib.import_module("drone_utils")
    ns = base_namespace(jax)
    ns.update(drone_params=drone_params, fori_loop=jax.lax.fori_loop,
              p_th_quantile_cdf_normal=drone_utils.p_th_quantile_cdf_normal)
    load_reference(os.path.join(REF, "drone", "drone_gaussian.py"), ns, overrides={"S": S})
    model = ns["Model"](S, 'gaussian', alpha)
    t = np.arange(S)[:, None]
    us = np.hstack([0.25 * np.cos(0.2 * t) + 0.1, 0.05 * np.sin(0.3 * t), 0.02 * np.cos(t)])
    out = dict(S=S, alpha=alpha, us=us, mass_variance=float(model.mass_variance),
               xs=npy(model.us_to_state_trajectory(jnp.array(us))),
               Sigmas=npy(model.us_to_covariance_trajectory(jnp.array(us))),
               b_dx=npy(model.b_dx(jnp.array(drone_params.x_init) + 0.3, jnp.array(us[3]))),
               b_dmass=npy(model.b_dmass(jnp.array(drone_params.x_init) + 0.3, jnp.array(us[3]))))
    np.savez_compressed(os.path.join(HERE, f"ref_gaussian_S{S}.npz"), **with_hashes(out, "drone/drone_gaussian.py"))
    sys.path.pop(0)


CASES = {"drone": ((20, 16), (50, 8)), "driving": ((20, 16), (40, 8)), "hopper": ((30, 30), (60, 24)),
         "gaussian": ((30, 0),)}


def main(which=None):
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: this generator only runs in the build container")
    jax = jax_standin.install()
    makers = {"drone": make_drone, "driving": make_driving, "hopper": make_hopper, "gaussian": make_gaussian}
    for name, mk in makers.items():
        if which and name not in which:
            continue
        for S, M in CASES[name]:
            t0 = time.time()
            mk(jax, S, M)
            print(f"ref_{name}_S{S}" + (f"_M{M}" if M else "") + f".npz  {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
