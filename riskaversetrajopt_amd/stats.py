"""Monte-Carlo risk statistics on the device (``rato_risk_stats``) and the
deterministic second stage of the sample mean (``rato_sum_partials``).

Mirrors ``monte_carlo_avar`` (drone_risk.py:663-695, driving.py:639-671,
hopper.py:926-958) and ``monte_carlo_var`` (drone_main_plot.py:640-652)."""
import numpy as np
import torch

from . import _lib

SATISFIED_THRESHOLD = 1e-6   # ``B_satisfied = max_constraint <= 1e-6`` (drone_risk.py:661)

_STAT_NAMES = ("var", "cvar", "frac_satisfied", "mean", "max", "count_satisfied", "tail_sum", "rank",
               "count_above_var", "count_at_var", "t_star")
N_STATS = len(_STAT_NAMES)


def _as_device_f32(Z, device=None):
    if isinstance(Z, torch.Tensor):
        if not Z.is_cuda:
            raise _lib.RatoError("risk statistics need a device tensor (no CPU fallback)")
        return Z.contiguous().float()
    return torch.as_tensor(np.asarray(Z, dtype=np.float32), device=device or 'cuda:0')


def new_workspace(M, device):
    """A rato_risk_stats workspace (device bytes), initialised once (rato_risk_stats_init); reusable for any number of
    stream-ordered calls on ONE stream at a time."""
    lib = _lib.load()
    ws = torch.empty(lib.rato_risk_stats_workspace_bytes(M), dtype=torch.uint8, device=device)
    with torch.cuda.device(ws.device):
        _lib.check(lib.rato_risk_stats_init(_lib.ptr(ws), ws.numel(), _lib.current_stream()), "rato_risk_stats_init")
    return ws


def sums_and_risk_stats_device(part, Z, alpha, thr=SATISFIED_THRESHOLD, scale=1.0, workspace=None, sums_out=None,
                               out=None, stream=None):
    """sum_partials(part) and risk_stats_device(Z) as ONE launch for M <= 1,048,576 (rato_sums_and_risk_stats)
    -> (sums fp64 (part.shape[1:]), stats fp64 [N_STATS])."""
    lib = _lib.load()
    _lib.require_f32_device(part, "part")
    Z = _as_device_f32(Z)
    M = Z.numel()
    if workspace is None:
        workspace = new_workspace(M, Z.device)
    if sums_out is None:
        sums_out = torch.empty(part.shape[1:], dtype=torch.float64, device=part.device)
    if out is None:
        out = torch.empty(N_STATS, dtype=torch.float64, device=Z.device)
    _lib.check(lib.rato_sums_and_risk_stats(_lib.ptr(part), part.shape[0], part[0].numel(), float(scale),
                                            _lib.ptr(sums_out), _lib.ptr(Z), M, float(alpha), float(thr),
                                            _lib.ptr(workspace), workspace.numel(), _lib.ptr(out),
                                            _lib.current_stream() if stream is None else stream),
               "rato_sums_and_risk_stats")
    return sums_out, out


def risk_stats_device(Z, alpha, thr=SATISFIED_THRESHOLD, workspace=None, out=None, stream=None):
    """Z: device tensor (M,) fp32 -> device tensor double[10] (see rato_saa.h).
    Asynchronous on the current stream (``stream``: an already looked-up ``_lib.current_stream()``)."""
    lib = _lib.load()
    Z = _as_device_f32(Z)
    M = Z.numel()
    if workspace is None:
        workspace = new_workspace(M, Z.device)
    if out is None:
        out = torch.empty(N_STATS, dtype=torch.float64, device=Z.device)
    _lib.check(lib.rato_risk_stats(_lib.ptr(Z), M, float(alpha), float(thr), _lib.ptr(workspace),
                                   workspace.numel(), _lib.ptr(out),
                                   _lib.current_stream() if stream is None else stream),
               "rato_risk_stats")
    return out


# ---- statistics in the producer's own launch (rato_saa.h: params.stats_*) -------------------------------------------
FUSED_MAX_M = 524288               # 64 statistics workgroups x 512 threads x 16 keys


STATS_IN_LAUNCH = 1                # RATO_STATS_IN_LAUNCH (params.stats_flags): rato_*_eval computes them in its own launch


def request_in_launch(params, workspace, out, alpha, thr=SATISFIED_THRESHOLD, flags=0):
    """Fill the ``stats_*`` fields of a ``rato_drone_params`` / ``rato_car_params``: the row-parallel linearize launch
    given these params then also leaves the ``rato_risk_stats`` record of the Z it produces in ``out`` (device
    double[N_STATS]) -- computed by extra workgroups at the end of its grid as soon as the last tile's Z has landed,
    while the Jacobian is still being stored."""
    if workspace is None or out is None:
        raise _lib.RatoError("statistics in the launch need an initialised workspace and a device record")
    params.stats_workspace = workspace.data_ptr()
    params.stats_out = out.data_ptr()
    params.stats_alpha = float(alpha)
    params.stats_thr = float(thr)
    params.stats_flags = int(flags)


def risk_stats_recover_device(Z, alpha, thr=SATISFIED_THRESHOLD, workspace=None, out=None, stream=None):
    """The same record by the launch-per-pass form on a re-initialised workspace (rato_risk_stats_recover): what to call
    when ``risk_stats_device`` came back NaN on finite input -- the one-launch forms give up, loudly, when the workgroups
    of their launch could not run together for seconds or the workspace was left unclean."""
    lib = _lib.load()
    Z = _as_device_f32(Z)
    M = Z.numel()
    if workspace is None:
        workspace = new_workspace(M, Z.device)
    if out is None:
        out = torch.empty(N_STATS, dtype=torch.float64, device=Z.device)
    _lib.check(lib.rato_risk_stats_recover(_lib.ptr(Z), M, float(alpha), float(thr), _lib.ptr(workspace),
                                           workspace.numel(), _lib.ptr(out),
                                           _lib.current_stream() if stream is None else stream),
               "rato_risk_stats_recover")
    return out


def risk_stats(Z, alpha, thr=SATISFIED_THRESHOLD, workspace=None):
    """-> dict with var, cvar, frac_satisfied, mean, max, ... (host floats).  A NaN record on finite input (a one-launch
    selection that gave up) is recovered through the launch-per-pass form before it is returned."""
    Z = _as_device_f32(Z)
    if workspace is None:
        workspace = new_workspace(Z.numel(), Z.device)
    out = risk_stats_device(Z, alpha, thr, workspace=workspace).cpu().numpy()
    if np.isnan(out[0]) and count_nonfinite(Z) == 0:
        out = risk_stats_recover_device(Z, alpha, thr, workspace=workspace).cpu().numpy()
    return dict(zip(_STAT_NAMES, out.tolist()))


def monte_carlo_var(Z_samples, alpha):
    """drone_main_plot.py:640-652."""
    return risk_stats(Z_samples, alpha)["var"]


def monte_carlo_avar(Z_samples, alpha):
    """drone_risk.py:663-695 (exact minimiser instead of the OSQP LP)."""
    return risk_stats(Z_samples, alpha)["cvar"]


def sum_partials(part, scale=1.0, out=None, stream=None):
    """part: device (nblocks, ...) fp32 (or fp64: the block sums of the CVaR-cut oracle) -> device double tensor of
    shape part.shape[1:] holding scale * sum over blocks (fixed order, fp64)."""
    lib = _lib.load()
    f64 = isinstance(part, torch.Tensor) and part.dtype == torch.float64
    if f64:
        if not (part.is_cuda and part.is_contiguous()):
            raise _lib.RatoError("part must be a contiguous device tensor (no CPU fallback)")
    else:
        _lib.require_f32_device(part, "part")
    nblocks = part.shape[0]
    ncols = part[0].numel()
    if out is None:
        out = torch.empty(part.shape[1:], dtype=torch.float64, device=part.device)
    fn = lib.rato_sum_partials_f64 if f64 else lib.rato_sum_partials
    _lib.check(fn(_lib.ptr(part), nblocks, ncols, float(scale), _lib.ptr(out),
                  _lib.current_stream() if stream is None else stream), "rato_sum_partials")
    return out


def count_nonfinite(x):
    """Number of NaN/Inf entries of a device fp32 tensor (one small kernel + a 4-byte readback)."""
    lib = _lib.load()
    _lib.require_f32_device(x, "x")
    cnt = torch.empty(1, dtype=torch.int32, device=x.device)
    _lib.check(lib.rato_count_nonfinite(_lib.ptr(x), x.numel(), _lib.ptr(cnt), _lib.current_stream()),
               "rato_count_nonfinite")
    return int(cnt.item())


def enqueue_nonfinite_count(what, *tensors, out=None):
    """Stream-ordered half of ``assert_finite``: counts the NaN/Inf entries of the given device fp32 tensors into ONE
    device counter (rato_count_nonfinite + _acc) and returns it WITHOUT reading it back (None if there is nothing to
    scan) -- a caller that reads other results back anyway copies the counter along and pays one synchronisation."""
    lib = _lib.load()
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return None
    cnt = out if out is not None else torch.empty(1, dtype=torch.int32, device=tensors[0].device)
    st = _lib.current_stream()
    for i, t in enumerate(tensors):
        _lib.require_f32_device(t, what)
        fn = lib.rato_count_nonfinite if i == 0 else lib.rato_count_nonfinite_acc
        _lib.check(fn(_lib.ptr(t), t.numel(), _lib.ptr(cnt), st), "rato_count_nonfinite")
    return cnt


def raise_if_nonfinite(what, bad):
    if bad:
        raise _lib.RatoNonFiniteError(f"{what}: {int(bad)} non-finite values in the device outputs (RATO_ENONFINITE)")


def assert_finite(what, *tensors):
    """Failure detection for the hot path (a distinct status on non-finite output; the reference only
    prints "[solve]: Problem infeasible.", drone_risk.py:458-459, and keeps iterating on NaNs): counts the NaN/Inf
    entries of the given device fp32 tensors into ONE device counter (rato_count_nonfinite + _acc), reads it back
    once and raises ``RatoNonFiniteError`` (status RATO_ENONFINITE) if it is not zero."""
    cnt = enqueue_nonfinite_count(what, *tensors)
    if cnt is not None:
        raise_if_nonfinite(what, int(cnt.item()))
