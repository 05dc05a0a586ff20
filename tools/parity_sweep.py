#!/usr/bin/env python3
"""How robust is "SCP subproblems of the device path match the fp64 path to 1e-5"?  Seed sweep of the comparison in
tests/test_gpu_scp.py::test_reduced_subproblems_device_vs_fp64_host_oracle: for every seed the device path (fp32
linearization, fp64 cut oracle) and the fp64 host path (tests/_host_cuts.py on the fp64 oracle's linearization) solve
every subproblem of the fp64 path's SCP sequence from the same iterate; prints the worst |du| / |dt_risk| per seed and
the iteration it occurred at.     python tools/parity_sweep.py [n_seeds] [M]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    S_arg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    # "round": the fp64 leg gets the samples rounded to fp32 (what the device holds), which separates the rounding of
    # the INPUTS from what the device arithmetic and its fp32 tables add
    round_inputs = len(sys.argv) > 4 and sys.argv[4] == "round"
    r32 = (lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)) if round_inputs else (lambda a: a)
    from oracle import drone as od, driving as ocar
    from riskaversetrajopt_amd import drone_risk, driving
    from tests._host_cuts import DroneReducedOracle, DrivingReducedOracle
    S = S_arg
    for system in ("drone", "driving"):
        worst_all = 0.0
        for seed in range(n_seeds):
            alpha = (0.05, 0.1, 0.2)[seed % 3]
            if system == "drone":
                DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(100 + seed), 'saa', M=M, S=S)
                o, d = od.Model(S, r32(DWs), r32(masses), r32(Q), 'saa', alpha), drone_risk.Model(S, DWs, masses, Q, 'saa', alpha)
                h, iters = DroneReducedOracle(o), 14
            else:
                samples = ocar.sample_uncertain_parameters(np.random.RandomState(100 + seed), M, 'saa', S)
                o = ocar.Model(*[r32(x) for x in samples], method='saa', alpha=alpha)
                d = driving.Model(M, 'saa', alpha, S=S, samples=samples)
                h, iters = DrivingReducedOracle(o), 9
            us = h.initial_guess_us_mat()
            du, dtr = [], []
            for k in range(iters):
                ud, td, _ = d.solve_reduced(us, k)
                uh, th, _ = h.solve_reduced(us, k)
                du.append(np.abs(ud - uh).max())
                dtr.append(abs(td - th))
                us = uh
            worst_all = max(worst_all, max(du))
            print(f"{system} M={M} seed={seed} alpha={alpha}: worst |du| {max(du):.1e} at iteration {int(np.argmax(du))}, "
                  f"last three {du[-3]:.1e} {du[-2]:.1e} {du[-1]:.1e}; worst |dt_risk| {max(dtr):.1e}")
        print(f"== {system}: worst |du| over {n_seeds} seeds {worst_all:.2e}")


if __name__ == "__main__":
    main()
