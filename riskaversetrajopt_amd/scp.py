"""SCP outer loop and the reference's timing protocol.

Mirrors the script-level driver of the reference (``drone_risk.py:495-540``,
``driving.py:467-529``, ``drone_times.py:509-550``): warm-up iterations, restart
from the initial guess, a FIXED number of iterations (no convergence test),
per-iteration wall-clock of "define" (linearize + assemble) and "solve" (host
QP) with cumulative times, and the L2 change of the controls."""
import time

import numpy as np


def L2_error_us(us_mat, us_mat_prev):
    error = np.mean(np.linalg.norm(us_mat - us_mat_prev, axis=-1))
    return error / np.mean(np.linalg.norm(us_mat, axis=-1))


def _sync():
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


def _finite_guard(model, check_finite):
    """Failure detection: every linearization of the run is scanned for NaN/Inf and a non-finite output raises
    ``RatoNonFiniteError`` (RATO_ENONFINITE) instead of the reference's print-and-continue (drone_risk.py:458-459)."""
    if check_finite is not None and hasattr(model, "check_finite"):
        model.check_finite = bool(check_finite)


def run_drone(model, num_scp_iters_max=60, warmup_iters=5, verbose=False, check_finite=True):
    """drone_risk.py:503-532: define once (scp_iter=2 pattern), ``warmup_iters`` throw-away
    iterations, restart, then a fixed number of update_problem/solve iterations.
    -> dict(us, t_risk, define_s, solve_s, cumulative_s, L2_error)"""
    _finite_guard(model, check_finite)
    us_prev = model.initial_guess_us_mat()
    model.define_problem(us_prev, verbose=False)
    for scp_iter in range(warmup_iters):
        model.update_problem(us_prev, scp_iter, verbose=False)
        us_prev, _ = model.solve(verbose=False)
    us_prev = model.initial_guess_us_mat()
    define_s, solve_s, err = [], [], []
    t_risk = None
    for scp_iter in range(num_scp_iters_max):
        _sync()
        t0 = time.perf_counter()
        model.update_problem(us_prev, scp_iter, verbose=False)
        _sync()
        t1 = time.perf_counter()
        us, t_risk = model.solve(verbose=False)
        t2 = time.perf_counter()
        define_s.append(t1 - t0)
        solve_s.append(t2 - t1)
        err.append(L2_error_us(us, us_prev))
        us_prev = us
        if verbose:
            print(f"scp {scp_iter:3d}  define {t1 - t0:.4f}s  solve {t2 - t1:.4f}s  L2 {err[-1]:.3e}")
    define_s, solve_s = np.array(define_s), np.array(solve_s)
    return {"us": us_prev, "t_risk": t_risk, "define_s": define_s, "solve_s": solve_s,
            "cumulative_s": np.cumsum(define_s + solve_s), "L2_error": np.array(err)}


def run_drone_reduced(model, num_scp_iters_max=60, verbose=False, check_finite=True, native_loop=None, tol=None):
    """The SCP loop (drone_risk.py:519-532; also used for driving, driving.py:486-513) with every subproblem solved through
    ``Model.solve_reduced`` (device CVaR oracle + host master QP) — the path that scales to M = 1e5.
    "define" = device linearization (+ its small read-backs), "solve" = cutting-plane loop.
    ``native_loop`` (default: whenever the Model offers it -- the drone, one GPU, table-free oracle): the whole loop as ONE
    library call with the per-iteration clocks taken natively (``Model.scp_run_native`` -> rato_scp_run_drone); False: the
    per-iteration Python loop below, which is also the checker of the native one (same iterates bit for bit) and what the
    native call hands back to when it meets a case only the Python loop recovers.
    ``tol``: stopping violation of the cutting-plane loops (default: ``solve_reduced``'s own, 1e-9)."""
    _finite_guard(model, check_finite)
    if hasattr(model, "_lib"):                 # a device Model: its master QP must be the native one (no silent NumPy leg)
        from . import dense_qp
        dense_qp.require_native()
    us_prev = model.initial_guess_us_mat()
    if native_loop is not False and not verbose and num_scp_iters_max > 0 and hasattr(model, "scp_run_native"):
        _sync()
        r = model.scp_run_native(us_prev, num_scp_iters_max, **({} if tol is None else {"tol": tol}))
        if r is not None:
            hist = r["us_hist"]
            prev = [np.asarray(us_prev, dtype=np.float64)] + list(hist[:-1])
            err = np.array([L2_error_us(u, p) for u, p in zip(hist, prev)])
            return {"us": hist[-1], "t_risk": float(r["t_risk"][-1]), "define_s": r["define_s"], "solve_s": r["solve_s"],
                    "cumulative_s": np.cumsum(r["define_s"] + r["solve_s"]), "L2_error": err, "cuts": r["cuts"],
                    "oracle_s": r["oracle_s"], "us_hist": hist, "loop": "native (rato_scp_run_drone)"}
        if native_loop is True:
            raise RuntimeError("run_drone_reduced(native_loop=True): the native SCP loop does not apply to this Model / "
                               "handed back")
        model._cut_solver = None                # (a handed-back run starts over with the per-iteration loop)
    define_s, solve_s, err, cuts, oracle_s, hist = [], [], [], [], [], []
    t_risk = None
    for scp_iter in range(num_scp_iters_max):
        _sync()
        t0 = time.perf_counter()
        us, t_risk, info = model.solve_reduced(us_prev, scp_iter, **({} if tol is None else {"tol": tol}))
        _sync()
        dt_total = time.perf_counter() - t0
        solve_s.append(info["oracle_s"] + info["master_s"])
        define_s.append(dt_total - solve_s[-1])
        oracle_s.append(info["oracle_s"])
        cuts.append(info["cuts"])
        err.append(L2_error_us(us, us_prev))
        hist.append(np.array(us, dtype=np.float64))
        us_prev = us
        if verbose:
            print(f"scp {scp_iter:3d}  define {define_s[-1]:.4f}s  solve {solve_s[-1]:.4f}s "
                  f"({info['cuts']} cuts, oracle {info['oracle_s']:.4f}s)  L2 {err[-1]:.3e}")
    define_s, solve_s = np.array(define_s), np.array(solve_s)
    return {"us": us_prev, "t_risk": t_risk, "define_s": define_s, "solve_s": solve_s,
            "cumulative_s": np.cumsum(define_s + solve_s), "L2_error": np.array(err), "cuts": np.array(cuts),
            "oracle_s": np.array(oracle_s), "us_hist": np.array(hist),
            "loop": "python (one define + one solve call per iteration)"}


def run_driving(model, num_scp_iters_max=15, verbose=False, check_finite=True):
    """driving.py:474-513: two warm-up solves (scp_iter 0 and 1), restart, then a fixed number of
    define_problem/solve iterations (define re-sets the solver up at iterations 0 and 1)."""
    _finite_guard(model, check_finite)
    us_prev = model.initial_guess_us_mat()
    model.define_problem(us_prev, verbose=False)
    us, _ = model.solve()
    model.define_problem(us, 1, verbose=False)
    us, _ = model.solve()
    us_prev = model.initial_guess_us_mat()
    define_s, solve_s, err = [], [], []
    t_risk = None
    for scp_iter in range(num_scp_iters_max):
        _sync()
        t0 = time.perf_counter()
        model.define_problem(us_prev, scp_iter, verbose=False)
        _sync()
        t1 = time.perf_counter()
        us, t_risk = model.solve()
        t2 = time.perf_counter()
        define_s.append(t1 - t0)
        solve_s.append(t2 - t1)
        err.append(L2_error_us(us, us_prev))
        us_prev = us
        if verbose:
            print(f"scp {scp_iter:3d}  define {t1 - t0:.4f}s  solve {t2 - t1:.4f}s  L2 {err[-1]:.3e}")
    define_s, solve_s = np.array(define_s), np.array(solve_s)
    return {"us": us_prev, "t_risk": t_risk, "define_s": define_s, "solve_s": solve_s,
            "cumulative_s": np.cumsum(define_s + solve_s), "L2_error": np.array(err)}


def save_results(path, *arrays):
    """The reference's result-file convention: several ``np.save`` calls appended to ONE file
    (``us`` then ``xs``: drone_risk.py:534-539; nine arrays: drone_main_plot.py:700-710)."""
    with open(path, 'wb') as f:
        for a in arrays:
            np.save(f, np.asarray(a))


def load_results(path, n):
    """Read back ``n`` arrays written by ``save_results`` (sequential ``np.load``, drone_risk.py:704-708)."""
    with open(path, 'rb') as f:
        return [np.load(f) for _ in range(n)]


def run_driving_reduced(model, num_scp_iters_max=15, verbose=False, check_finite=True):
    """driving.py:486-513 with ``Model.solve_reduced`` subproblems (same loop as run_drone_reduced)."""
    return run_drone_reduced(model, num_scp_iters_max=num_scp_iters_max, verbose=verbose, check_finite=check_finite)


def monte_carlo_report(mc_model, us_list, alpha, verbose=False):
    """The out-of-sample validation block of the reference's scripts (drone_risk.py:697-725, driving.py:672-700):
    every solution in ``us_list`` (the SAA repeats of one alpha) is evaluated on the Monte-Carlo model's fresh
    samples (M = 10000 there) -- fraction of samples that satisfy the constraints, AVaR_alpha of the max constraint
    value, control cost -- and the mean / median over the repeats are reported.  All repeats in ONE library call where the
    model has the batched entry point (drone, driving: ``Model.eval_batch_device``), one rollout kernel + one exact selection
    per solution otherwise; all on the device.  -> dict of per-solution arrays and the aggregates."""
    frac, avar, var, cost = [], [], [], []
    batched = getattr(mc_model, "eval_batch_device", None)
    if batched is not None and len(us_list) > 1 and getattr(mc_model, "_dW", None) is not None:
        # all repeats of this alpha in ONE call (rato_*_eval_batch: one rollout launch over tiles x K, one launch of K
        # exact selections); row k is what the per-solution call below gives for solution k, to the bit
        from . import stats
        _, rec = batched(np.stack([np.asarray(us) for us in us_list]), alpha=alpha)
        rec = rec.cpu().numpy()
        names = stats._STAT_NAMES
        for k, us in enumerate(us_list):
            st = dict(zip(names, rec[k].tolist()))
            if np.isnan(st["var"]):                    # (a selection that gave up: the recovering per-solution path)
                st = mc_model.monte_carlo_statistics(us, alpha=alpha)
            frac.append(st["frac_satisfied"])
            avar.append(st["cvar"])
            var.append(st["var"])
            cost.append(mc_model.monte_carlo_cost(us))
            if verbose:
                print("B_satisfied_vec =", frac[-1])
        us_list = []
    for us in us_list:
        st = mc_model.monte_carlo_statistics(us, alpha=alpha)
        frac.append(st["frac_satisfied"])
        avar.append(st["cvar"])
        var.append(st["var"])
        cost.append(mc_model.monte_carlo_cost(us))
        if verbose:
            print("B_satisfied_vec =", frac[-1])
    out = {"frac_satisfied": np.array(frac), "avar": np.array(avar), "var": np.array(var), "cost": np.array(cost)}
    for k in ("frac_satisfied", "avar", "cost"):
        out[k + "_mean"], out[k + "_median"] = float(np.mean(out[k])), float(np.median(out[k]))
    if verbose:
        print("percentage safe (mean) =", out["frac_satisfied_mean"])
        print("avar (mean) =", out["avar_mean"])
        print("cost (mean) =", out["cost_mean"])
        print("percentage safe (median) =", out["frac_satisfied_median"])
        print("avar (median) =", out["avar_median"])
        print("cost (median) =", out["cost_median"])
    return out
