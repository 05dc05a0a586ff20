#!/usr/bin/env python3
"""SCP wall-clock with the reference's timing protocol (drone_times.py:509-550, driving.py:482-529):
per-iteration "define" (device linearization + sparse assembly) and "solve" (host QP) times and the
cumulative time, medians over the iterations; plus the define-only time of the device path at large M
(where the host QP — 1.5e7 rows at M = 1e5 — is out of reach of any host solver, the reference's included).

    python scp_bench.py --system drone --M 50 --S 20 --iters 15
"""
import argparse
import json
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="drone", choices=["drone", "driving"])
    ap.add_argument("--seed", type=int, default=0, help="seed of the device sampler (bench.py's scp block uses 7)")
    ap.add_argument("--M", type=int, default=50)
    ap.add_argument("--S", type=int, default=20)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--reduced", action="store_true",
                    help="solve every subproblem in (u, slack) with device CVaR cuts (scales to M = 1e6)")
    ap.add_argument("--tol", type=float, default=None, help="--reduced: violation at which the cutting-plane loop stops")
    ap.add_argument("--define-only-M", type=int, default=100000,
                    help="also time linearize+means+statistics alone at this M (0 = skip)")
    args = ap.parse_args()
    import torch
    from riskaversetrajopt_amd import scp, dist as rdist
    np.random.seed(0)
    # under torchrun (--reduced only): --M samples PER GPU, the cutting-plane oracle runs across the ranks
    rank, world, local = rdist.init_from_env() if args.reduced else (0, 1, 0)
    if world > 1:
        torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if args.system == "drone" and args.reduced:
        from riskaversetrajopt_amd import drone_risk, drone_utils
        dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(args.M, args.S, seed=1000 * rank + args.seed, device=dev)
        model = drone_risk.Model.from_device(args.S, dW, mass, Qsym, 'saa', args.alpha, M=args.M)
        if world > 1:
            model.shard()
        model.solve_reduced(model.initial_guess_us_mat(), 2)             # warm-up (allocations, first launches)
        t_all = time.perf_counter()
        out = scp.run_drone_reduced(model, num_scp_iters_max=args.iters, verbose=False, tol=args.tol)   # (one GPU: rato_scp_run_drone)
        if rank == 0:
            for k in range(args.iters):
                print(f"scp {k:3d}  define {out['define_s'][k]:.4f}s  solve {out['solve_s'][k]:.4f}s "
                      f"({out['cuts'][k]} cuts, oracle {out['oracle_s'][k]:.4f}s)  L2 {out['L2_error'][k]:.3e}")
        line = {"system": "drone", "mode": "reduced (u, slack) problem, device CVaR cuts + host master QP",
                "loop": out.get("loop"),
                "M": args.M, "M_total": args.M * world, "n_gpus": world, "S": args.S, "alpha": args.alpha,
                "iters": args.iters,
                "define_median_s": float(np.median(out["define_s"])), "solve_median_s": float(np.median(out["solve_s"])),
                "oracle_median_s": float(np.median(out["oracle_s"])), "cuts_median": float(np.median(out["cuts"])),
                "cuts_max": int(out["cuts"].max()), "cumulative_s": float(out["cumulative_s"][-1]),
                "wall_s": time.perf_counter() - t_all, "L2_error_last": float(out["L2_error"][-1]),
                "define_total_s": float(np.sum(out["define_s"])), "oracle_total_s": float(np.sum(out["oracle_s"])),
                "master_total_s": float(np.sum(out["solve_s"]) - np.sum(out["oracle_s"])), "cuts_total": int(out["cuts"].sum())}
        st = model.monte_carlo_statistics(out["us"], alpha=args.alpha)          # this rank's shard
        line["in_sample"] = {k: st[k] for k in ("var", "cvar", "frac_satisfied")}
        if rank == 0:
            print(json.dumps(line))
        return
    if args.system == "driving" and args.reduced:
        from riskaversetrajopt_amd import driving
        dW, x0, ws, wr = driving.sample_uncertain_parameters_device(args.M, args.S, seed=1000 * rank + args.seed, device=dev)
        model = driving.Model.from_device(args.S, dW, x0, ws, wr, 'saa', args.alpha)
        if world > 1:
            model.shard()
        model.solve_reduced(model.initial_guess_us_mat(), 1)             # warm-up
        t_all = time.perf_counter()
        out = scp.run_driving_reduced(model, num_scp_iters_max=args.iters, verbose=(rank == 0))
        line = {"system": "driving", "mode": "reduced (u, slack) problem, device CVaR cuts + host master QP",
                "M": args.M, "S": args.S, "alpha": args.alpha, "iters": args.iters,
                "define_median_s": float(np.median(out["define_s"])), "solve_median_s": float(np.median(out["solve_s"])),
                "oracle_median_s": float(np.median(out["oracle_s"])), "cuts_median": float(np.median(out["cuts"])),
                "cuts_max": int(out["cuts"].max()), "cumulative_s": float(out["cumulative_s"][-1]),
                "wall_s": time.perf_counter() - t_all, "L2_error_last": float(out["L2_error"][-1])}
        st = model.monte_carlo_statistics(out["us"], alpha=args.alpha)
        line["in_sample"] = {k: st[k] for k in ("var", "cvar", "frac_satisfied")}
        print(json.dumps(line))
        return
    if args.system == "drone":
        from riskaversetrajopt_amd import drone_risk, drone_utils
        DWs, masses, obs_Qs = drone_utils.sample_uncertain_parameters('saa', M=args.M, S=args.S)
        model = drone_risk.Model(args.S, DWs, masses, obs_Qs, 'saa', args.alpha)
        out = scp.run_drone(model, num_scp_iters_max=args.iters, warmup_iters=2)
    else:
        from riskaversetrajopt_amd import driving
        model = driving.Model(args.M, 'saa', args.alpha, S=args.S)
        out = scp.run_driving(model, num_scp_iters_max=args.iters)
    line = {"system": args.system, "M": args.M, "S": args.S, "alpha": args.alpha, "iters": args.iters,
            "define_median_s": float(np.median(out["define_s"])), "solve_median_s": float(np.median(out["solve_s"])),
            "cumulative_s": float(out["cumulative_s"][-1]), "L2_error_last": float(out["L2_error"][-1]),
            "qp_rows": int(model.A.shape[0]), "qp_cols": int(model.A.shape[1]), "qp_nnz": int(model.A.nnz),
            "qp_status": model.res.info.status}
    if args.define_only_M and args.system == "drone":
        from riskaversetrajopt_amd import drone_risk, drone_utils, stats
        M = args.define_only_M
        dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, 50, dt=1.0)
        big = drone_risk.Model.from_device(50, dW, mass, Qsym, 'saa', args.alpha, M=M)
        us = big._us_device(np.tile([0.3, 0.05, 0.0], (50, 1)))
        r = big.linearize_device(us)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r = big.linearize_device(us, out=r)
            stats.risk_stats_device(r["Z"], args.alpha)
        torch.cuda.synchronize()
        line["define_only"] = {"M": M, "S": 50, "seconds_per_iteration": (time.perf_counter() - t0) / 20,
                               "what": "device linearize + sample means + VaR/CVaR (outputs stay in HBM)"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
    import torch.distributed as _dist
    if _dist.is_available() and _dist.is_initialized():
        _dist.destroy_process_group()
