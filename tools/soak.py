"""Soak test: thousands of launches of the row-parallel kernels (LDS progress polling, work queues) at varied M / S,
checking that every launch completes and is bitwise reproducible.  usage: python tools/soak.py [seconds]"""
import sys, time, numpy as np, torch, faulthandler
faulthandler.dump_traceback_later(600, exit=True)
sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils, driving
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    S = int(rng.choice([2, 3, 7, 20, 33, 50, 64, 100, 126]))
    M = int(rng.choice([1, 5, 63, 64, 65, 300, 4097, 20000, 100000, 250000]))
    if S * M > 2e7:
        continue
    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=int(rng.randint(1 << 30)), device=dev)
    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
    us = (rng.randn(S, 3) * 0.5).astype(np.float32)
    a = d.linearize_device(us, want_A22=bool(rng.randint(2)))
    valid = lambda r: {"G": drone_risk.untile(r["G"], M), "W": r["W"], "g_up": r["g_up"], "Z": r["Z"], "sums": r["sums"]}
    ref = {k: v.clone() for k, v in valid(a).items()}      # (lanes >= M of the last tile are never written)
    for _ in range(int(rng.randint(2, 6))):
        a = d.linearize_device(us, out=a, want_A22=a["_A22"] is not None)
        torch.cuda.synchronize()
        assert all(bool((v == ref[k]).all()) for k, v in valid(a).items()), ("drone", S, M)
    g = d.linearize_generators_device(us)
    g2 = d.linearize_generators_device(us)
    assert bool((g["g_up"] == g2["g_up"]).all()) and bool((g["sums"] == g2["sums"]).all())
    n += 1
    if S <= 100:
        Sc = max(S, 2)
        dWc, x0, ws, wr = driving.sample_uncertain_parameters_device(M, Sc, seed=int(rng.randint(1 << 30)), device=dev)
        c = driving.Model.from_device(Sc, dWc, x0, ws, wr, 'saa', 0.05)
        usc = (rng.randn(Sc, 2) * 0.2).astype(np.float32)
        b = c.linearize_device(usc)
        validc = lambda r: {"G": driving.untile(r["G"], M), "g_up": r["g_up"], "Z": r["Z"]}
        refc = {k: v.clone() for k, v in validc(b).items()}
        for _ in range(3):
            b = c.linearize_device(usc, out={k: b[k] for k in ("G", "g_up", "Z", "final_du", "final_rhs")})
            torch.cuda.synchronize()
            assert all(bool((v == refc[k]).all()) for k, v in validc(b).items()), ("driving", Sc, M)
        n += 1
    del d, dW, a, ref
print("soak ok: %d configurations" % n)
