"""Store-only and copy bandwidth ceilings of the box (torch fill_/copy_ on 3 GB), to put the write-bound row kernel's
achieved GB/s next to what a plain streaming store reaches.  usage: python tools/store_peak.py"""
import torch
dev = torch.device("cuda:0")
n = 3 * (1 << 28)                      # 3 Gi bytes of fp32 / 4
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for _ in range(3):
    ms = t(lambda: x.fill_(1.0))
    print("fill  %.3f ms  %.0f GB/s (store only)" % (ms, 4 * n / ms / 1e6))
    ms = t(lambda: x.zero_())
    print("zero  %.3f ms  %.0f GB/s (memset)" % (ms, 4 * n / ms / 1e6))
    ms = t(lambda: y.copy_(x))
    print("copy  %.3f ms  %.0f GB/s (read + write)" % (ms, 8 * n / ms / 1e6))
    ms = t(lambda: x.sum())
    print("sum   %.3f ms  %.0f GB/s (read only)" % (ms, 4 * n / ms / 1e6))
