import torch, sys, time
sys.path.insert(0, "/root/repo")
import fab_torch_amd as fa
from fab_torch_amd import _ops
DEV = "cuda"
ops = _ops.load()
for (D, K, nodes, B) in [(32, 10, 10, 1024), (32, 4, 8, 300), (6, 3, 40, 70), (16, 3, 16, 257), (2, 2, 64, 33), (32, 2, 4, 1152)]:
    torch.manual_seed(D * 7 + K)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    eps = torch.randn(B, D, device=DEV)
    ops.set_option(_ops.OPT_TILE_SHAPE, 16)
    x16, lq16 = flow.native_sample(eps)
    ops.set_option(_ops.OPT_TILE_SHAPE, 0)
    x4, lq4 = flow.native_sample(eps)
    lq_chk = flow.log_prob(x4)
    ex = float((x16 - x4).abs().max() / x16.abs().max())
    el = float((lq16 - lq4).abs().max() / lq16.abs().max())
    ec = float((lq_chk - lq4).abs().max() / lq4.abs().max())
    print(f"D={D} K={K} W={D*nodes} B={B}: x rel err {ex:.2e}, log q rel err {el:.2e}, vs density of x {ec:.2e}, equal={torch.equal(x16, x4)}")
flow = None
