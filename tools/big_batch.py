"""One-off check of 64-bit indexing at a batch that fills a large part of the 288 GB HBM: drone linearize at M = 8e6,
S = 50 (Phi alone is 78 GB), last samples compared with the fp64 oracle.  usage: python tools/big_batch.py [M]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import drone as od
from riskaversetrajopt_amd import drone_risk, drone_utils
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
S = 50
dev = torch.device("cuda:0")
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=3, device=dev)
d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
t = np.arange(S)[:, None]
us = np.hstack([0.6*np.cos(0.3*t)+0.3, 0.15*np.sin(0.5*t)+0.02, 0.05*np.cos(t)])*(20/S)
r = d.linearize_device(us)
torch.cuda.synchronize()
t0 = time.perf_counter(); r = d.linearize_device(us, out=r); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("M=%d  G %.1f GB  linearize %.1f ms  (%.0f GB/s)" % (M, r["G"].numel()*4/1e9, dt*1e3, (r["G"].numel()+r["_W"].numel()+r["_g_up"].numel())*4/dt/1e9))
idx = np.array([0, 1, 63, 64, M//2, M-65, M-64, M-2, M-1])
ti = torch.as_tensor(idx, device=dev)
# oracle inputs for those samples from the device-side SoA tensors
DWs = np.zeros((len(idx), S, 6)); DWs[:, :, 3:6] = dW[:, :, ti].permute(2, 0, 1).double().cpu().numpy()
masses = mass[ti].double().cpu().numpy()
Q = Qsym[:, :, ti].double().cpu().numpy()            # (3 obs, 3, n): (Q00, Q01+Q10, Q11)
obs_Qs = np.zeros((len(idx), 3, 3, 3))
obs_Qs[:, :, 0, 0] = Q[:, 0].T; obs_Qs[:, :, 1, 1] = Q[:, 2].T; obs_Qs[:, :, 0, 1] = Q[:, 1].T
o = od.Model(S, DWs, masses, obs_Qs, 'saa', 0.1)
o.dt = d.dt
_, _, _, gdu_o, gup_o = o.get_all_constraints_coeffs(us)
tile = r["tile"]
Phi = torch.stack([r["G"][m // tile, :, :, m % tile] for m in idx.tolist()], dim=-1)      # (n_pairs, 2, n)
t_of = torch.as_tensor(np.concatenate([np.full(tt, tt) for tt in range(1, S)]), device=dev)
Wt = r["W"][:, :, :, ti].permute(1, 2, 0, 3)[t_of]                                          # (n_pairs, 2, 3, n)
gdu = d.expand_g_obs_du(Phi[:, :, None, :] * Wt)
err = np.abs(gdu - gdu_o).max() / np.abs(gdu_o).max()
gerr = np.abs(r["g_up"][:, :, ti].permute(2, 0, 1).cpu().numpy() - gup_o).max()
print("sampled Jacobian rel err %.2e   g_up abs err %.2e" % (err, gerr))
assert err < 1e-4 and gerr < 5e-4
print("ok")
