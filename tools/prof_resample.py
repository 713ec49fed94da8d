"""Profiling driver for the resample path at large N (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lw = torch.randn(N, device=dev, generator=g) * 3
u = torch.rand(N, dtype=torch.float64, device=dev, generator=g)
x = torch.randn(N, 8, device=dev, generator=g)
for _ in range(5):
    idx = fa.systematic_indices(lw, u0=0.3)
    idx2 = fa.multinomial_indices(lw, u=u)
    y = fa.gather_rows(x, idx)
    e = fa.ess_and_log_z(lw)
torch.cuda.synchronize()
print("done", N)
