"""Phase times inside rs_small (diagnostic build -DRATO_RS_DIAG loaded through RATO_SAA_LIB): wall-clock stamps
written behind the statistics.  usage: RATO_SAA_LIB=tools/_build/librato_rsdiag.so python tools/rs_small_phases.py"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from riskaversetrajopt_amd import stats
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
names = ["zero h", "load+pass1", "find1", "pass2", "find2", "pass3", "find3", "tail", "reduce+write"]
for M in (1000, 10000):
    for name, Z in (("clustered", 0.9 + 0.05 * torch.randn(M, generator=g, device=dev)),
                    ("straddling 0", -1.0 + 1.2 * torch.randn(M, generator=g, device=dev))):
        ws = stats.new_workspace(M, dev)
        out = torch.zeros(32, dtype=torch.float64, device=dev)
        big = torch.empty(1 << 27, device=dev)
        acc = []
        for rep in range(8):
            big.fill_(1.0)                       # evict L2 like a producer kernel would
            stats.risk_stats_device(Z, 0.1, workspace=ws, out=out)
            torch.cuda.synchronize()
            st = out.cpu().numpy()[16:25]
            acc.append(np.diff(st) / 100.0)      # us
        a = np.median(np.array(acc), axis=0)
        print("M=%6d %-13s total %.2f us: " % (M, name, a.sum()) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names[1:], a)))
