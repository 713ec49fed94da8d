"""Dev check of the second-generation 4 / 8-chain tiles (flow_r4w.h) through fabhip_flow_log_prob: density + gradient against
the 16-chain kernel on the same inputs, 4- vs 8-chain bit-equality, HIP-event time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fab_torch_amd as fa  # noqa: E402
from fab_torch_amd import _ops  # noqa: E402

dev = torch.device("cuda", 0)


def run(flow, x, shape):
    with _ops.option(_ops.OPT_TILE_SHAPE, shape):
        lq, g = flow.log_prob_and_grad(x)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for i in range(10):
            flow.log_prob_and_grad(x)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[5]
    return lq, g, ms


def flows():
    yield "headline D=32 W=320 K=10", bench.build_flow_state(0).to(dev).requires_grad_(False)
    torch.manual_seed(3)
    f = fa.RealNVP(6, 8, 40)
    with torch.no_grad():
        for p in f.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    yield "D=6 W=240 K=8", f.to(dev).requires_grad_(False)
    torch.manual_seed(4)
    f = fa.RealNVP(32, 4, 8)      # W = 256
    with torch.no_grad():
        for p in f.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    yield "D=32 W=256 K=4", f.to(dev).requires_grad_(False)


for name, flow in flows():
    for B in (1024, 2048, 1000):
        x = torch.randn(B, flow.dim, device=dev) * 1.2
        lq16, g16, t16 = run(flow, x, 16)
        lq4, g4, t4 = run(flow, x, 4)
        lq8, g8, t8 = run(flow, x, 8)
        sc = float(lq16.abs().max())
        print(f"{name} B={B}: |lq4-lq16|/max {float((lq4 - lq16).abs().max()) / sc:.2e}  |g4-g16|/max "
              f"{float((g4 - g16).abs().max() / g16.abs().max()):.2e}  4==8 bits: {torch.equal(lq4, lq8) and torch.equal(g4, g8)}  "
              f"finite {bool(torch.isfinite(g4).all())}  ms 16/4/8: {t16:.3f} {t4:.3f} {t8:.3f}")
