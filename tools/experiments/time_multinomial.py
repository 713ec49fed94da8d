import torch, time, sys
sys.path.insert(0, "/root/repo")
from fab_torch_amd import _ops
ops = _ops.load()
N = 1 << 26
g = torch.Generator(device="cuda").manual_seed(0)
lw = torch.randn(N, device="cuda", generator=g)
u = torch.rand(N, device="cuda", dtype=torch.float64, generator=g)
for _ in range(3): ops.resample_multinomial(lw, u)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): idx = ops.resample_multinomial(lw, u)
torch.cuda.synchronize(); print("multinomial e2e ms", (time.perf_counter() - t0) / 10 * 1e3, int(idx.sum()) % 1000003)
