"""Build container only (run as a SUBPROCESS by tests/test_jax_standin.py: installing the stand-in patches torch.Tensor).

(A) every operation ``jax_standin`` maps, against NumPy's semantics on the same data -- the stand-in carries the
    reference's text in ``make_reference_golden.py``, so a wrong ``reshape(order='F')`` or ``.at[].set`` would be baked
    into the fixtures;
(B) where /root/reference exists: the Jacobians the reference's own ``get_all_constraints_coeffs`` returns through the
    stand-in's ``jacfwd`` against CENTRAL FINITE DIFFERENCES of the reference's own forward text (rollout + constraint
    functions) -- ties the stand-in's differentiation to the reference's forward arithmetic independently of oracle/."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import jax_standin  # noqa: E402

jax = jax_standin.install()
jnp = jax.numpy
import torch  # noqa: E402


def same(a, b, tol=0.0):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.max(np.abs(a - b), initial=0.0) <= tol, np.max(np.abs(a - b))


rng = np.random.RandomState(0)
x = rng.randn(2, 3, 4)
# ---- reshape, both orders, every rank change the reference uses (and some it does not)
for shape in [(6, 4), (4, 6), (24,), (3, 8), (2, 12), (4, 3, 2)]:
    for order in "CF":
        same(jnp.reshape(x, shape, order), np.reshape(x, shape, order))
v = rng.randn(12)
same(jnp.reshape(v, (3, 4), 'F').T, np.reshape(v, (3, 4), 'F').T)          # drone_risk.py:95-100 (vec -> mat)
same(jnp.reshape(jnp.array(x[0]), (12,), 'C'), np.reshape(x[0], (12,), 'C'))    # :102-106 (mat -> vec)
same(jnp.reshape(jnp.array(x[0]), 12), np.reshape(x[0], 12))
# ---- .at[idx].set: ints, slices, tuples, negative indices, broadcast values; the source stays untouched
a = jnp.array(x)
for idx, val in [((0,), 7.0), ((slice(None), 1), np.arange(4.0)), ((1, slice(1, 3), slice(None, None, 2)), -1.0),
                 ((-1, -1, -1), 3.5), ((slice(None), slice(None), 0), rng.randn(2, 3)), ((0, 2), rng.randn(4))]:
    ref = x.copy()
    ref[idx] = val
    same(a.at[idx].set(val), ref)
same(a, x)
z = jnp.zeros((5, 3))
ref = np.zeros((5, 3)); ref[1:, :2] = 4.0
same(z.at[1:, :2].set(4.0), ref)
# ---- repeat / concatenate / hstack (1-D and 2-D) / vstack
same(jnp.repeat(x[0], 3, axis=0), np.repeat(x[0], 3, axis=0))
same(jnp.repeat(x[0], 2, axis=1), np.repeat(x[0], 2, axis=1))
same(jnp.repeat(x[0], 2), np.repeat(x[0], 2))
same(jnp.repeat(x[0][None], 5, axis=0), np.repeat(x[0][None], 5, axis=0))
same(jnp.concatenate((x[0], x[1]), axis=0), np.concatenate((x[0], x[1]), axis=0))
same(jnp.concatenate((x[0], x[1]), axis=-1), np.concatenate((x[0], x[1]), axis=-1))
same(jnp.hstack((v, v[:3])), np.hstack((v, v[:3])))
same(jnp.hstack((x[0], x[1][:, :2])), np.hstack((x[0], x[1][:, :2])))
same(jnp.vstack((x[0], x[1])), np.vstack((x[0], x[1])))
same(jnp.vstack((v, v)), np.vstack((v, v)))
# ---- reductions and elementwise
for ax in (None, 0, 1, -1):
    same(jnp.mean(x[0], axis=ax), np.mean(x[0], axis=ax), 1e-14)
    same(jnp.sum(x[0], axis=ax), np.sum(x[0], axis=ax), 1e-14)
    same(jnp.max(x[0], axis=ax), np.max(x[0], axis=ax))
same(np.max(jnp.array(x[0])), np.max(x[0]))                       # np.max ON a stand-in array (drone_risk.py:660)
same(jnp.array(x[0]).max(axis=1), x[0].max(axis=1))
same(jnp.linalg.norm(x[0], axis=-1), np.linalg.norm(x[0], axis=-1), 1e-15)
same(jnp.linalg.norm(v), np.linalg.norm(v), 1e-15)
same(jnp.dot(v, v), np.dot(v, v), 1e-14)
same(jnp.dot(x[0], x[1].T), np.dot(x[0], x[1].T), 1e-14)
same(jnp.diag(v[:4]), np.diag(v[:4]))
same(jnp.maximum(x[0], 0.1), np.maximum(x[0], 0.1))
same(jnp.abs(x[0]), np.abs(x[0])); same(jnp.sqrt(np.abs(x[0])), np.sqrt(np.abs(x[0])))
same(jnp.sin(x[0]), np.sin(x[0]), 1e-15); same(jnp.cos(x[0]), np.cos(x[0]), 1e-15)
same(jnp.median(v), np.median(v))
# ---- a NumPy array on the LEFT of an operator with a stand-in array on the right
t = jnp.array(x[0])
same(x[1] * t, x[1] * x[0]); same(x[1] + t, x[1] + x[0]); same(x[1] - t, x[1] - x[0]); same(x[1] / t, x[1] / x[0], 1e-15)
same(x[1].T[:3] @ jnp.array(x[0][:, :2]), x[1].T[:3] @ x[0][:, :2], 1e-14)
# ---- vmap (leading axis, tuple outputs), jacfwd / jacrev / hessian / grad with argnums
f = lambda p, q: (p * q.sum(), jnp.sin(p) @ q)
o0, o1 = jax.vmap(f)(x[0], x[1])
same(o0, np.stack([x[0][i] * x[1][i].sum() for i in range(3)]), 1e-15)
same(o1, np.stack([np.sin(x[0][i]) @ x[1][i] for i in range(3)]), 1e-15)
A = rng.randn(4, 4)
g = lambda p: jnp.dot(jnp.array(A), p * p)
same(jax.jacfwd(g)(v[:4]), A * (2 * v[:4])[None, :], 1e-14)
same(jax.jacrev(g)(v[:4]), A * (2 * v[:4])[None, :], 1e-14)
h = lambda p: jnp.sum(jnp.dot(jnp.array(A), p * p) * p)
H = jax.hessian(h)(v[:4])
eps = 1e-5
fd = np.zeros((4, 4))
hn = lambda p: float(np.sum((A @ (p * p)) * p))
for i in range(4):
    for j in range(4):
        ei, ej = np.eye(4)[i] * eps, np.eye(4)[j] * eps
        fd[i, j] = (hn(v[:4] + ei + ej) - hn(v[:4] + ei - ej) - hn(v[:4] - ei + ej) + hn(v[:4] - ei - ej)) / (4 * eps * eps)
same(H, fd, 1e-4)
k2 = lambda p, q: jnp.sum(p * q * q)
same(jax.grad(k2, argnums=1)(v[:4], v[4:8]), 2 * v[:4] * v[4:8], 1e-14)
same(jax.jacfwd(lambda p, q: p * q * q, argnums=1)(v[:4], v[4:8]), np.diag(2 * v[:4] * v[4:8]), 1e-14)
print("standin ops ok")

# ---------------------------------------------------------------- (B) the reference's own text
REF = os.environ.get("RATO_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    print("reference absent: part B skipped")
    sys.exit(0)
import importlib  # noqa: E402
import make_reference_golden as mk  # noqa: E402


def central_fd(fwd, us, h=1e-6):
    us = np.asarray(us, dtype=np.float64)
    base = [np.asarray(mk.npy(o)) for o in fwd(jnp.array(us))]
    jac = [np.zeros(b.shape + us.shape) for b in base]
    for idx in np.ndindex(*us.shape):
        up, dn = us.copy(), us.copy()
        up[idx] += h
        dn[idx] -= h
        for j, (a, b) in enumerate(zip(fwd(jnp.array(up)), fwd(jnp.array(dn)))):
            jac[j][(Ellipsis,) + idx] = (mk.npy(a) - mk.npy(b)) / (2 * h)
    return jac


S, M = 8, 2
sys.path.insert(0, os.path.join(REF, "drone"))
drone_params = importlib.import_module("drone_params")
drone_utils = importlib.import_module("drone_utils")
ns = mk.base_namespace(jax)
ns.update(drone_params=drone_params, sample_uncertain_parameters=drone_utils.sample_uncertain_parameters)
mk.load_reference(os.path.join(REF, "drone", "drone_risk.py"), ns, overrides={"S": S, "M": M})
np.random.seed(1)
DWs, masses, obs_Qs = ns["sample_uncertain_parameters"]('saa', M=M, S=S)
model = ns["Model"](S, jnp.array(DWs), jnp.array(masses), jnp.array(obs_Qs), 'saa', 0.1)
us = np.asarray(mk.drone_us(S, "graze"))
i = 1
fwd = lambda u: (model.final_constraints(model.us_to_state_trajectory(u, model.masses[i], model.DWs[i])),
                 model.obstacle_avoidance_constraints(model.us_to_state_trajectory(u, model.masses[i], model.DWs[i]),
                                                      model.obs_Qs[i]))
fdu, _, _, gdu, _ = model.get_all_constraints_coeffs(jnp.array(us), model.masses[i], model.DWs[i], model.obs_Qs[i])
j_final, j_obs = central_fd(fwd, us)
e1 = np.abs(mk.npy(fdu) - j_final.reshape(mk.npy(fdu).shape)).max() / max(1e-30, np.abs(j_final).max())
e2 = np.abs(mk.npy(gdu) - j_obs.reshape(mk.npy(gdu).shape)).max() / max(1e-30, np.abs(j_obs).max())
print(f"drone: reference jacfwd (stand-in) vs central differences of the reference's forward text: {e1:.1e} {e2:.1e}")
assert e1 < 1e-7 and e2 < 1e-7
sys.path.pop(0)

sys.path.insert(0, os.path.join(REF, "car"))
driving_params = importlib.import_module("driving_params")
ns = mk.base_namespace(jax)
ns.update(driving_params=driving_params)
mk.load_reference(os.path.join(REF, "car", "driving.py"), ns, overrides={"S": S, "M": M, "dt": driving_params.T / S})
np.random.seed(1)
model = ns["Model"](M, 'saa', 0.05)
us = np.asarray(mk.car_us(S, "swerve"))
args = (model.states_init[i], model.omegas_speed[i], model.omegas_repulsive[i], model.DWs[i])
fwd = lambda u: (model.final_constraints(model.us_to_state_trajectory(u, *args)),
                 -model.separation_distances_at_all_times(model.us_to_state_trajectory(u, *args)))
fdu, _, _, gdu, _ = model.get_all_constraints_coeffs(jnp.array(us), *args)
j_final, j_obs = central_fd(fwd, us)
e1 = np.abs(mk.npy(fdu) - j_final.reshape(mk.npy(fdu).shape)).max() / max(1e-30, np.abs(j_final).max())
e2 = np.abs(mk.npy(gdu) - j_obs.reshape(mk.npy(gdu).shape)).max() / max(1e-30, np.abs(j_obs).max())
print(f"driving: reference jacfwd (stand-in) vs central differences of the reference's forward text: {e1:.1e} {e2:.1e}")
assert e1 < 1e-7 and e2 < 1e-7
print("reference text ok")
