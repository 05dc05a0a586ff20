import sys, numpy as np, torch, faulthandler
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, '.')
from oracle import drone as od
from riskaversetrajopt_amd import drone_risk
S, M = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(0)
DWs, masses, Q = od.sample_uncertain_parameters(rng, 'saa', M=M, S=S)
d = drone_risk.Model(S, DWs, masses, Q)
t = np.arange(S)[:, None]
us = np.hstack([0.6*np.cos(0.3*t)+0.3, 0.15*np.sin(0.5*t)+0.02, 0.05*np.cos(t)])*(20/S)
print("plan", d.linearize_plan(M, d._mass.numel()), flush=True)
r = d.linearize_device(us, cols_per_thread=int(sys.argv[3]))
print("launched", flush=True)
torch.cuda.synchronize()
print("synced", r["du_sum"].sum().item(), r["Z"][:3].tolist(), flush=True)
