"""Measured floor of the SCP-iterate parity (north star: 1e-5): device-linearized SCP vs fp64-oracle-linearized SCP,
same host QP (polished on both legs), per iteration.  usage: python tools/scp_tol.py"""
import sys
import numpy as np
sys.path.insert(0, '.')
from riskaversetrajopt_amd import scp, drone_risk, driving
from oracle import drone as od, driving as ocar
from tests._oracle_qp import DroneOracleQP, DrivingOracleQP


def iterates(model, run, n, **kw):
    us_list = []
    orig = model.solve

    def solve(*a, **k):
        us, t = orig(*a, **k)
        us_list.append(np.array(us))
        return us, t
    model.solve = solve
    out = run(model, num_scp_iters_max=n, **kw)
    return np.array(us_list[-n:]), out


for S, M, alpha, seed in ((20, 30, 0.2, 0), (20, 50, 0.1, 1)):
    DWs, masses, Q = od.sample_uncertain_parameters(np.random.RandomState(seed), 'saa', M=M, S=S)
    o = od.Model(S, DWs, masses, Q, 'saa', alpha)
    d = drone_risk.Model(S, DWs, masses, Q, 'saa', alpha)
    a, ra = iterates(DroneOracleQP(o), scp.run_drone, 25, warmup_iters=1)
    b, rb = iterates(d, scp.run_drone, 25, warmup_iters=1)
    diff = np.abs(a - b).reshape(25, -1).max(1)
    print(f"drone S={S} M={M}: max|du| per iteration (u_max=10): first {diff[:3]}, last 5 {diff[-5:]}, max {diff.max():.2e}; "
          f"relative to u_max {diff.max() / 10:.2e}; polish ok oracle/device: "
          f"{ra.get('polish', '?')}/{rb.get('polish', '?')}")
for S, M, alpha, seed in ((20, 16, 0.1, 0), (20, 40, 0.05, 2)):
    x0, ws, wr, DWs = ocar.sample_uncertain_parameters(np.random.RandomState(seed), M, 'saa', S)
    o = ocar.Model(x0, ws, wr, DWs, 'saa', alpha)
    d = driving.Model(M, 'saa', alpha, S=S, samples=(x0, ws, wr, DWs))
    a, _ = iterates(DrivingOracleQP(o), scp.run_driving, 10)
    b, _ = iterates(d, scp.run_driving, 10)
    diff = np.abs(a - b).reshape(10, -1).max(1)
    print(f"driving S={S} M={M}: max|du| per iteration (u_max=100): {diff}, max {diff.max():.2e}; relative to u_max {diff.max() / 100:.2e}")
