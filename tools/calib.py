import torch, time
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for gb in (0.3, 3.0, 12.0):
    n=int(gb*1e9/4)
    x=torch.empty(n, device='cuda'); y=torch.empty(n, device='cuda')
    t=bench(lambda: x.fill_(1.0)); print(f"fill {gb} GB: {t:.3f} ms -> {gb/t*1e3:.0f} GB/s")
    t=bench(lambda: x.zero_()); print(f"zero_ {gb} GB: {t:.3f} ms -> {gb/t*1e3:.0f} GB/s")
    t=bench(lambda: y.copy_(x)); print(f"copy {gb} GB: {t:.3f} ms -> {2*gb/t*1e3:.0f} GB/s (r+w)")
    t=bench(lambda: torch.mul(x, 2.0, out=y)); print(f"mul {gb} GB: {t:.3f} ms -> {2*gb/t*1e3:.0f} GB/s (r+w)")
    del x,y
