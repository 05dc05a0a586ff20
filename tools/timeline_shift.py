"""Experiment: does the odd/even tile asymmetry of drone_linearize_rows_kernel follow the ADDRESS parity of the
per-sample arrays (dW, mass, Qsym, g_up, Z: tile i covers the 256-B chunk i of every row) or the workgroup's place?
All per-sample base pointers are shifted by `shift` floats (row stride ld + 64) through the C ABI directly.
usage (RATO_SAA_LIB = a -DRATO_DIAG=4 build): python tools/timeline_shift.py [shift floats]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import _lib, drone_risk, drone_utils
lib = _lib.load()
M, S = 100000, 50
shift = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
dW0, mass0, Q0 = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device=dev)
ld0 = mass0.numel()
ld = ld0 + 64
def padded(t):          # [..][ld0] -> storage [..][ld], data starting `shift` floats into every row
    big = torch.zeros(t.shape[:-1] + (ld,), dtype=torch.float32, device=dev)
    big[..., shift:shift + ld0] = t
    return big
dW, mass, Q = padded(dW0), padded(mass0), padded(Q0)
mass[mass == 0] = 32.0
d = drone_risk.Model.from_device(S, dW0, mass0, Q0, 'saa', 0.1, M=M)
p = d._params(M, ld)
t = np.arange(S)[:, None]
us = d._us_device(np.hstack([0.6*np.cos(0.3*t)+0.3, 0.15*np.sin(0.5*t)+0.02, 0.05*np.cos(t)])*(20/S))
nt = (M + 63) // 64
G = torch.empty((nt, S * (S - 1) // 2, 2, 3, 64), dtype=torch.float32, device=dev)
g_up = torch.zeros((3, S, ld), dtype=torch.float32, device=dev)
Z = torch.zeros(ld, dtype=torch.float32, device=dev)
part = torch.empty((nt, 6 * S + 6), dtype=torch.float32, device=dev)
off = lambda x: C.c_void_p(x.data_ptr() + 4 * shift)
for _ in range(4):
    rc = lib.rato_drone_linearize(C.byref(p), _lib.ptr(us), off(dW), off(mass), off(Q), _lib.ptr(G), None, None, off(g_up),
                                  off(Z), _lib.ptr(part), -1, 1, _lib.current_stream())
    assert rc == 0
torch.cuda.synchronize()
tl = np.ascontiguousarray(part.cpu().numpy()[:, :8]).view(np.uint64).astype(np.float64)
tl = (tl - tl[:, 0].min()) * 1e-2
start, p0, p1, end = tl.T
print("shift %d floats: kernel span %.1f us" % (shift, end.max()))
for x in range(2):
    sel = np.arange(nt) % 2 == x
    print("  tiles = %d mod 2: lifetime mean %.1f us  staging %.1f  last end %.1f" % (x, (end - start)[sel].mean(), (p0 - start)[sel].mean(), end[sel].max()))
