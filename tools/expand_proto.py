#!/usr/bin/env python3
"""A/B record for the "address-ordered expansion" form of the drone products Jacobian (tools/expand_proto.hip; VERDICT r2
item 4): kernel 1 = the library's generators-only linearization (A22, W, g_up, Z, sums), kernel 2 = the prototype
expansion kernel that writes the packed products Jacobian from those tables in address order.  Prints, on one board and
alternating: the row-parallel products kernel of the library (one launch), kernel 1, kernel 2 for several chunk sizes /
workgroup sizes, and the sum; checks kernel 2's Jacobian against the library's.   Run on the GPU box:
    python tools/expand_proto.py [M] [S]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riskaversetrajopt_amd import _lib, drone_risk, drone_utils   # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    so = "/tmp/expand_proto.so"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
                    "-o", so, os.path.join(ROOT, "tools", "expand_proto.hip")], check=True)
    lib = C.CDLL(so)
    lib.expand_proto.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int,
                                 C.c_float, C.c_float, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
    dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=7)
    d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
    t = np.arange(S)[:, None]
    us = d._us_device(np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S))
    ref = d.linearize_device(us, factored=False)                      # the library's products kernel
    gen = d.linearize_generators_device(us)
    G2 = _lib.packed_buffer(tuple(ref["G"].shape), d.device)
    stride = G2.stride(0)
    p = d._params(M, mass.numel())
    st = _lib.current_stream()

    def expand(CH, NW):
        rc = lib.expand_proto(_lib.ptr(gen["_A22"]), 3, _lib.ptr(gen["_W"]), _lib.ptr(mass), _lib.ptr(G2), M,
                              mass.numel(), S, p.dt, p.kp, CH, NW, stride, st)
        assert rc == 0, rc

    G2.zero_()
    expand(11, 4)
    torch.cuda.synchronize()
    a, b = drone_risk.untile(ref["G"], M), drone_risk.untile(G2, M)
    scale = a.abs().max().item()
    err = (a - b).abs().max().item()
    print(f"M={M} S={S}: prototype vs library products Jacobian: max |diff| {err:.3e} of scale {scale:.3e}")
    assert err <= 2e-5 * scale

    def timed(fn, n=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    jac_bytes = M * 4 * 3 * S * (S - 1)
    out_rows = d.linearize_device(us, factored=False)
    for rep in range(3):
        t_rows = timed(lambda: d.linearize_device(us, factored=False, out=out_rows, reduce=False))
        t_gen = timed(lambda: d.linearize_generators_device(us, out=gen))
        line = f"rep {rep}: rows kernel (one launch) {t_rows:.4f} ms | kernel 1 (generators) {t_gen:.4f} ms | kernel 2:"
        for CH, NW in ((11, 4), (11, 8), (22, 4), (43, 4), (6, 4), (11, 1)):
            t2 = timed(lambda: expand(CH, NW))
            line += f"  CH={CH} NW={NW}: {t2:.4f} ms ({jac_bytes / t2 / 1e6:.0f} GB/s)"
        print(line)
    t_fill = timed(lambda: G2.fill_(1.0))
    print(f"torch.fill_ of the same buffer ({G2.numel() * 4 / 1e9:.2f} GB incl. padding): {t_fill:.4f} ms")


if __name__ == "__main__":
    main()
