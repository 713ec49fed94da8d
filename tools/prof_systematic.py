"""profiling target: the fused systematic resampler at N = 2^26 (a few calls)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fab_torch_amd import _ops
ops = _ops.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
g = torch.Generator(device="cuda").manual_seed(0)
lw = torch.randn(N, device="cuda", generator=g) * 3
for _ in range(4):
    idx = ops.resample_systematic(lw, 0.3, N)
torch.cuda.synchronize()
