"""A minimal stand-in for the parts of ``jax`` the reference's hot path touches, backed by torch fp64.

WHY.  The reference (/root/reference, pure Python) delegates its array type, batching and
differentiation to ``jax`` (``jit``, ``vmap``, ``jacfwd``, ``jacrev``, ``hessian``; unpinned in its
requirements.txt), which is not installed in the build container and cannot be fetched.  To pin the
oracle against the REFERENCE'S OWN TEXT anyway, ``make_reference_golden.py`` extracts the reference's
``class Model`` (and the Monte-Carlo closures) from the files under /root/reference at run time and
executes them unmodified against this module: every arithmetic statement that runs is the reference's,
only the array library underneath is torch (fp64) instead of XLA (fp64), and derivatives come from
``torch.func`` forward/reverse-mode autodiff instead of jax's.

This is test-infrastructure for the build container only: it never ships with the product, never runs on
the GPU box, and it is NOT jax — a later reader may still call the parity "pinned through a stand-in".

Semantics reproduced (only what the reference uses): immutable arrays with ``x.at[idx].set(v)``,
``jnp.reshape(x, shape, order)`` with 'C'/'F', ``jnp.repeat/concatenate/hstack/vstack/mean/...``,
``vmap`` over the leading axis of every argument (tuple outputs stacked), ``jacfwd``/``jacrev``/``hessian``
with respect to argument 0 (tuple outputs differentiated leaf by leaf), ``jit`` = identity.
"""
import sys
import types

import numpy as np
import torch

F64 = torch.float64


def _t(x):
    """anything array-like -> fp64 torch tensor (tensors pass through untouched so autodiff wrappers survive)."""
    if isinstance(x, torch.Tensor):
        return x if x.dtype == F64 or not x.dtype.is_floating_point else x.to(F64)
    if isinstance(x, (list, tuple)) and any(isinstance(e, torch.Tensor) for e in x):
        return torch.stack([_t(e) for e in x])
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


# ---- x.at[idx].set(v): functional update (out of place, so that it composes with torch.func) -----------
class _AtIndex:
    def __init__(self, x, idx):
        self.x, self.idx = x, idx

    def set(self, v):
        x = self.x
        lin = torch.arange(x.numel()).reshape(x.shape)[self.idx]        # positions that are overwritten
        v = _t(v).expand(lin.shape) if lin.dim() else _t(v).reshape(())
        flat = x.reshape(-1).index_put((lin.reshape(-1),), v.reshape(-1))
        return flat.reshape(x.shape)


class _At:
    def __init__(self, x):
        self.x = x

    def __getitem__(self, idx):
        return _AtIndex(self.x, idx)


def _install_tensor_methods():
    """jax arrays have .at and .to_py(); NumPy arrays on the LEFT of an operator must defer to the tensor."""
    torch.Tensor.at = property(lambda self: _At(self))
    torch.Tensor.to_py = lambda self: self.detach().cpu().numpy()

    def coerce(name, fn):
        orig = getattr(torch.Tensor, name)

        def wrapped(self, other):
            if isinstance(other, np.ndarray):
                return fn(_t(other), self)
            return orig(self, other)
        setattr(torch.Tensor, name, wrapped)

    orig_max = torch.Tensor.max

    def tmax(self, *a, axis=None, out=None, **k):       # np.max(x) on a jax array -> x.max(axis=None, out=None)
        if axis is not None:
            return orig_max(self, axis, **k).values
        return orig_max(self, *a, **k)
    torch.Tensor.max = tmax

    coerce("__rmul__", lambda a, b: a * b)
    coerce("__radd__", lambda a, b: a + b)
    coerce("__rsub__", lambda a, b: a - b)
    coerce("__rtruediv__", lambda a, b: a / b)
    coerce("__rmatmul__", lambda a, b: a @ b)


# ---- jax.numpy ------------------------------------------------------------------------------------------
def _reshape(x, shape, order='C'):
    x = _t(x)
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    if order == 'C':
        return x.reshape(shape)
    if order != 'F':
        raise ValueError(order)
    nd_in, nd_out = x.dim(), len(shape)
    xt = x.permute(*reversed(range(nd_in))) if nd_in > 1 else x
    y = xt.reshape(tuple(reversed(shape)))
    return y.permute(*reversed(range(nd_out))) if nd_out > 1 else y


def _axis_kw(axis):
    return {} if axis is None else {"dim": axis}


def _cat(seq, axis=0):
    return torch.cat([torch.atleast_1d(_t(s)) for s in seq], dim=axis)


def _make_jnp():
    jnp = types.ModuleType("jax.numpy")
    jnp.newaxis = None
    jnp.inf = float("inf")
    jnp.pi = np.pi
    jnp.zeros = lambda shape, dtype=None: torch.zeros(shape, dtype=F64)
    jnp.ones = lambda shape, dtype=None: torch.ones(shape, dtype=F64)
    jnp.eye = lambda n: torch.eye(n, dtype=F64)
    jnp.array = lambda x, dtype=None: _t(x).clone() if isinstance(x, torch.Tensor) else _t(x)
    jnp.asarray = jnp.array
    jnp.reshape = _reshape
    jnp.repeat = lambda x, n, axis=None: torch.repeat_interleave(_t(x), n, **_axis_kw(axis))
    jnp.concatenate = lambda seq, axis=0: _cat(seq, axis)
    jnp.hstack = lambda seq: _cat(seq, 0) if _t(seq[0]).dim() <= 1 else _cat(seq, 1)
    jnp.vstack = lambda seq: torch.cat([torch.atleast_2d(_t(s)) for s in seq], dim=0)
    jnp.mean = lambda x, axis=None: torch.mean(_t(x), **_axis_kw(axis))
    jnp.sum = lambda x, axis=None: torch.sum(_t(x), **_axis_kw(axis))
    jnp.max = lambda x, axis=None: torch.max(_t(x)) if axis is None else torch.max(_t(x), dim=axis).values
    jnp.median = lambda x, axis=None: torch.as_tensor(np.median(_t(x).detach().numpy(), axis=axis))
    jnp.abs = lambda x: torch.abs(_t(x))
    jnp.sqrt = lambda x: torch.sqrt(_t(x))
    jnp.sin = lambda x: torch.sin(_t(x))
    jnp.cos = lambda x: torch.cos(_t(x))
    jnp.dot = lambda a, b: torch.dot(_t(a), _t(b)) if _t(a).dim() == 1 and _t(b).dim() == 1 else _t(a) @ _t(b)
    jnp.diag = lambda x: torch.diag(_t(x))
    jnp.maximum = lambda a, b: torch.maximum(_t(a), _t(b))
    jnp.zeros_like = lambda x: torch.zeros_like(_t(x))
    linalg = types.ModuleType("jax.numpy.linalg")
    linalg.norm = lambda x, axis=None: torch.linalg.norm(_t(x)) if axis is None else torch.linalg.norm(_t(x), dim=axis)
    jnp.linalg = linalg
    return jnp


# ---- transformations ------------------------------------------------------------------------------------
def jit(fn=None, static_argnums=None, **_):
    if fn is None:
        return lambda f: f
    return fn


def _stack_tree(outs):
    first = outs[0]
    if isinstance(first, (tuple, list)):
        return tuple(_stack_tree([o[k] for o in outs]) for k in range(len(first)))
    return torch.stack([_t(o) if not isinstance(o, (bool, np.bool_)) else torch.as_tensor(bool(o)) for o in outs])


def vmap(fn):
    """Leading axis of every positional argument, mapped by a plain loop (the batch sizes here are tiny)."""
    def mapped(*args):
        args = [_t(a) if not isinstance(a, torch.Tensor) else a for a in args]
        n = args[0].shape[0]
        for a in args:
            if a.shape[0] != n:
                raise ValueError("vmap: mismatched leading axes %s" % ([tuple(a.shape) for a in args],))
        return _stack_tree([fn(*[a[i] for a in args]) for i in range(n)])
    return mapped


def _wrt0(transform):
    def outer(fn, argnums=0):
        def run(*args):
            args = list(args)
            x = _t(args[argnums])

            def only(z):
                a = list(args)
                a[argnums] = z
                return fn(*a)
            return transform(only)(x)
        return run
    return outer


def fori_loop(lower, upper, body, init):
    """jax.lax.fori_loop: a plain loop (the carried arrays are immutable, so this is the same computation)."""
    val = init
    for i in range(lower, upper):
        val = body(i, val)
    return val


jacfwd = _wrt0(torch.func.jacfwd)
jacrev = _wrt0(torch.func.jacrev)
hessian = _wrt0(torch.func.hessian)
grad = _wrt0(torch.func.grad)


def install():
    """Register the stand-in as ``jax`` / ``jax.numpy`` / ``jax.config`` / ``jax.scipy.stats`` in sys.modules."""
    if "jax" in sys.modules and not getattr(sys.modules["jax"], "_RATO_STANDIN", False):
        raise RuntimeError("a real jax is importable: use it instead of the stand-in")
    _install_tensor_methods()
    jax = types.ModuleType("jax")
    jax._RATO_STANDIN = True
    jax.numpy = _make_jnp()
    jax.jit, jax.vmap, jax.jacfwd, jax.jacrev, jax.hessian, jax.grad = jit, vmap, jacfwd, jacrev, hessian, grad
    cfg_mod = types.ModuleType("jax.config")

    class _Config:
        def update(self, *a, **k):
            pass
    cfg_mod.config = _Config()
    jax.config = cfg_mod
    jscipy = types.ModuleType("jax.scipy")
    jstats = types.ModuleType("jax.scipy.stats")
    import scipy.stats
    jstats.norm = scipy.stats.norm
    jscipy.stats = jstats
    jax.scipy = jscipy
    lax = types.ModuleType("jax.lax")
    lax.fori_loop = fori_loop
    jax.lax = lax
    sys.modules.update({"jax": jax, "jax.numpy": jax.numpy, "jax.config": cfg_mod, "jax.scipy": jscipy,
                        "jax.scipy.stats": jstats, "jax.lax": lax})
    return jax
