import torch, numpy as np, sys
sys.path.insert(0, '.')
from riskaversetrajopt_amd import stats
print("current stream handle:", torch.cuda.current_stream().cuda_stream)
for sync in (True, False):
    for scale in (1.0, 0.5):
        part = torch.randn(33, 33, device="cuda") * 100
        if sync:
            torch.cuda.synchronize()
        out = stats.sum_partials(part, scale=scale).cpu().numpy()
        ref = scale * part.double().sum(0).cpu().numpy()
        print("sync", sync, "scale", scale, "maxerr", np.abs(out - ref).max())
part = torch.randn(4, 306, device="cuda") * 100
o1 = stats.sum_partials(part, scale=0.5)
torch.cuda.synchronize()
print("direct", (o1.cpu().numpy() - 0.5 * part.double().sum(0).cpu().numpy()).max())
