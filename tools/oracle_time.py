#!/usr/bin/env python3
"""Per-launch time of the table-free cut-oracle kernels against the batch size (HIP events, 50 launches back to back):
is the rowmax kernel bound by instruction issue (time ~ waves per SIMD) or by the latency of one wave's chain?
    python tools/oracle_time.py [S]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riskaversetrajopt_amd import _lib, drone_risk, drone_utils, driving, stats   # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
lib = _lib.load()
for M in (12500, 25000, 50000, 65536, 100000, 131072, 200000, 400000, 1000000):
    dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=1, device=dev)
    d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
    p = d._params(M, mass.numel())
    uk = torch.as_tensor(np.asarray(d.initial_guess_us_mat(), dtype=np.float64), device=dev).contiguous()
    x = torch.zeros_like(uk) + 0.01
    m = torch.empty(M, dtype=torch.float32, device=dev)
    a = torch.empty(M, dtype=torch.int32, device=dev)
    st = torch.zeros(stats.N_STATS + 2 * (S - 1) + 1, dtype=torch.float64, device=dev)
    part = torch.zeros(((M + 255) // 256, 2 * (S - 1) + 1), dtype=torch.float64, device=dev)

    def rowmax():
        _lib.check(lib.rato_drone_rowmax_rollout(C.byref(p), _lib.ptr(uk), _lib.ptr(dW), _lib.ptr(mass), _lib.ptr(Qsym),
                                                 _lib.ptr(x), _lib.ptr(m), _lib.ptr(a), _lib.current_stream()), "rowmax")

    def tail():
        _lib.check(lib.rato_drone_tail_rows_rollout(C.byref(p), _lib.ptr(uk), _lib.ptr(dW), _lib.ptr(mass), _lib.ptr(Qsym),
                                                    _lib.ptr(m), _lib.ptr(a), _lib.ptr(st), st.numel(), None, 1, 0.1 * M,
                                                    _lib.ptr(part), _lib.current_stream()), "tail")
    rowmax()
    stats.risk_stats_device(m, 0.1, out=st[:stats.N_STATS])
    out = []
    for fn in (rowmax, tail):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print("M %8d  waves %6d (%.2f per SIMD)  rowmax %7.1f us  tail rows %7.1f us" % (M, (M + 63) // 64, (M + 63) // 64 / 1024, out[0], out[1]))
