#!/usr/bin/env python3
"""How many workgroups of car_linearize_rows_kernel store at the same time?  Needs a -DRATO_CDIAG=6 build of the library
(RATO_SAA_LIB=/path/to/lib.so): every workgroup leaves, per tile, the 100 MHz tick at which it took the tile, at which its
rollout wave finished (from here on all eight waves sweep and store rows) and at which the tile's last row was stored.
Printed: the phase lengths, and over the launch the number of workgroups in each phase (2 us bins) -- a write path shared
by everybody is used best when that number is steady.       python tools/car_timeline.py [M] [S]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riskaversetrajopt_amd import driving   # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=5)
d = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)
t = np.arange(S)[:, None]
us = np.hstack([0.4 * np.cos(0.3 * t) + 0.1, 0.03 * np.sin(0.5 * t) + 0.004]) * (20.0 / S)
for _ in range(4):
    r = d.linearize_device(us)
torch.cuda.synchronize()
n_wg = int(os.environ.get("NWG", 512))
raw = r["g_up"].reshape(-1)[:n_wg * 64].cpu().numpy().reshape(n_wg, 64).astype(np.int64)
n = raw[:, 0]
ev = []
for b in range(n_wg):
    for k in range(min(int(n[b]), 20)):
        ev.append((b, raw[b, 1 + 3 * k], raw[b, 2 + 3 * k], raw[b, 3 + 3 * k]))
ev = np.array(ev, dtype=np.int64)
t0 = ev[:, 1].min()
ev[:, 1:] = (ev[:, 1:] - t0) % (1 << 24)
start, rows, end = (ev[:, i] * 1e-2 for i in (1, 2, 3))        # microseconds
print("M %d S %d: %d workgroups, %d tiles recorded (%.2f per workgroup); launch span %.1f us" % (M, S, n_wg, len(ev), n.mean(), end.max()))
print("per tile (us): staging + rollout %.1f (p10 %.1f p90 %.1f)   rows after the rollout %.1f (p10 %.1f p90 %.1f)" % (
    (rows - start).mean(), *np.percentile(rows - start, [10, 90]), (end - rows).mean(), *np.percentile(end - rows, [10, 90])))
first = np.array([start[ev[:, 0] == b].min() for b in range(n_wg) if (ev[:, 0] == b).any()])
print("first tile taken at (us): min %.1f median %.1f max %.1f" % (first.min(), np.median(first), first.max()))
grid = np.arange(0.0, end.max() + 2.0, 2.0)
print("t (us): workgroups before the rows phase / in the rows phase / done or between tiles")
line = []
in_rows_all = []
for g in grid:
    a = int(((start <= g) & (rows > g)).sum())
    b = int(((rows <= g) & (end > g)).sum())
    in_rows_all.append(b)
    line.append("%3.0f:%3d/%3d/%3d" % (g, a, b, n_wg - a - b))
for i in range(0, len(line), 8):
    print("   ".join(line[i:i + 8]))
in_rows_all = np.array(in_rows_all)
body = in_rows_all[(grid > 20) & (grid < end.max() - 30)]
if len(body):
    print("workgroups in the rows phase between 20 us and 30 us before the end: mean %.0f  min %d  max %d  (std %.0f)" % (
        body.mean(), body.min(), body.max(), body.std()))
    print("bytes per tile %.0f KB -> a steady %.0f storing workgroups at %.1f us per rows phase is %.2f TB/s" % (
        780 * 2 * 4 * 64 / 1e3, body.mean(), (end - rows).mean(), body.mean() * 780 * 2 * 4 * 64 / ((end - rows).mean() * 1e-6) / 1e12))
