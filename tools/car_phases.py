#!/usr/bin/env python3
"""Where does a tile's time go in car_linearize_rows_kernel?  Needs a -DRATO_CDIAG=4 (or 5: without the Jacobian stores)
build of the library (RATO_SAA_LIB=/path/to/lib.so): every workgroup leaves its phase times (100 MHz ticks, thread 0)
at the front of g_up.       python tools/car_phases.py [M] [S]"""
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
for _ in range(3):
    r = d.linearize_device(us)
torch.cuda.synchronize()
n_wg = int(os.environ.get("NWG", 512))
raw = r["g_up"].reshape(-1)[:n_wg * 8].cpu().numpy().reshape(n_wg, 8)
tiles, stage, roll, rows, nxt, total = (raw[:, i] for i in range(6))
us_ = 1e-2                                    # ticks -> microseconds
print("M %d S %d: %d workgroups, tiles per workgroup %.2f (min %d max %d)" % (M, S, n_wg, tiles.mean(), tiles.min(), tiles.max()))
print("per workgroup, mean (us): total %.1f  | staging %.1f  rollout %.1f  rows after the rollout %.1f  barrier + queue fetch %.1f" % (
    total.mean() * us_, stage.mean() * us_, roll.mean() * us_, rows.mean() * us_, nxt.mean() * us_))
print("per tile, mean (us):      total %.1f  | staging %.2f  rollout %.2f  rows after the rollout %.2f  barrier + queue fetch %.2f" % (
    (total / tiles).mean() * us_, (stage / tiles).mean() * us_, (roll / tiles).mean() * us_, (rows / tiles).mean() * us_,
    (nxt / np.maximum(tiles - 1, 1)).mean() * us_))
print("unaccounted (prologue, final barrier): %.1f us" % ((total - stage - roll - rows - nxt).mean() * us_))
