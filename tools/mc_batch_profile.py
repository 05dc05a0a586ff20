#!/usr/bin/env python3
"""The reference's Monte-Carlo report form on the device: 120 control sequences per call on one validation batch
(rato_drone_eval_batch / rato_car_eval_batch at BASELINE C2 / C3), 50 calls each -- run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations behind `configs.C*_eval.batched` of the bench line."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riskaversetrajopt_amd import drone_risk, drone_utils, driving   # noqa: E402

K, reps = 120, 50
t = lambda S: np.arange(S)[:, None]
for system, M, S in (("drone", 10000, 50), ("driving", 10000, 40)):
    if system == "drone":
        dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7)
        d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
        us = np.hstack([0.6 * np.cos(0.3 * t(S)) + 0.3, 0.15 * np.sin(0.5 * t(S)) + 0.02, 0.05 * np.cos(t(S))]) * (20.0 / S)
    else:
        dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=7)
        d = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)
        us = np.hstack([0.4 * np.cos(0.4 * t(S)) - 0.2, 0.05 * np.sin(0.35 * t(S)) + 0.01]) * (20.0 / S)
    usb = torch.as_tensor(np.stack([us * (1.0 + 0.001 * k) for k in range(K)]), dtype=torch.float32, device=d.device)
    bufs = {}
    for _ in range(5):
        d.eval_batch_device(usb, out=bufs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        d.eval_batch_device(usb, out=bufs)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / reps * 1e6
    print(f"{system} M={M} S={S}: {K} sequences per call: {per:.1f} us per call = {per / K:.2f} us per sequence")
