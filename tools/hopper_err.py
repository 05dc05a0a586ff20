import sys, numpy as np
sys.path.insert(0, '.')
from oracle import hopper as oh
from riskaversetrajopt_amd import hopper
S, M = 60, 50000
fields = oh.sample_friction_fields(np.random.RandomState(1), M)
o = oh.Model(*fields, method='saa', alpha=0.2, S=S)
d = hopper.Model(M, 'saa', 0.2, S=S, fields=fields)
rng = np.random.RandomState(5)
C = 40
for pmax in (0.2, 3.0):
    px = np.linspace(-pmax, pmax, C); fz = 32 + rng.randn(C); forces = np.stack([0.08*fz + 0.3*rng.randn(C), fz], 1)
    h_o, dfz_o, dpx_o = o.slip_partials(px, forces)
    h, dfz, dpx = d.slip_partials(px, forces)
    _, Z = d.no_slip_constraints_verification(px, forces)
    print("pmax", pmax, "max|dh|", np.abs(h-h_o).max(), "max|dmu|", np.abs(dfz-dfz_o).max(), "max|ddpx|", np.abs(dpx-dpx_o).max(), "max|dZ|", np.abs(Z - h_o.max(1)).max())
