"""Shader clock by rato_device_clock_probe, idle and right after load.  usage: python tools/clock_probe.py"""
import sys, torch
sys.path.insert(0, '.')
from riskaversetrajopt_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
p = torch.zeros(3, dtype=torch.float64, device=dev)
x = torch.empty(1 << 28, device=dev)
for us in (20, 200, 2000):
    for load in (False, True):
        if load:
            for _ in range(20):
                x.fill_(1.0)
        _lib.check(lib.rato_device_clock_probe(_lib.ptr(p), us, _lib.current_stream()), "probe")
        torch.cuda.synchronize()
        print(us, "us", "after load" if load else "idle", "%.0f MHz over %.1f us" % tuple(p.cpu().numpy()[:2]))
