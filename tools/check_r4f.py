"""Dev tool: the fused-stage 4-chain kernels (FABHIP_OPT_R4_STREAM = 2) against the round-3 stream kernels (1), the 16-chain
tiles and the float64 oracle: flow sample (x, log q) and density + gradient at the sampled points."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _ops
from oracle import flow as oflow

dev = torch.device("cuda", 0)
D, K, nodes, B = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 10, 10, 1024)))
torch.manual_seed(D + K)
flow = fa.RealNVP(D, K, nodes).to(dev).requires_grad_(False)
with torch.no_grad():
    for p in flow.parameters():
        if p.dim() == 2 and p.shape[0] != p.shape[1]:
            p.add_(0.03 * torch.randn_like(p))
nf = oflow.make_realnvp(D, K, nodes).double()
nf.load_state_dict({k: v.double().cpu() for k, v in flow._nf_model.state_dict().items()})
g = torch.Generator(device=dev).manual_seed(5)
eps = torch.randn(B, D, device=dev, generator=g)
xo, lqo = (t.detach() for t in nf.sample_eps(eps.double().cpu()))
ops = _ops.load()
res = {}
for mode, shape in ((2, 4), (1, 4), (2, 16)):
    ops.set_option(_ops.OPT_R4_STREAM, mode)
    ops.set_option(_ops.OPT_TILE_SHAPE, shape)
    x, lq = flow.native_sample(eps)
    res[(mode, shape)] = (x.double().cpu(), lq.double().cpu())
ops.set_option(_ops.OPT_R4_STREAM, 2)
ops.set_option(_ops.OPT_TILE_SHAPE, 0)
sc = float(xo.abs().max())
for k, (x, lq) in res.items():
    ex = (x - xo).abs()
    print(k, "x err vs f64 oracle: max %.3e (scale %.2f) at row %d; log q err max %.3e (scale %.1f)" %
          (float(ex.max()), sc, int(ex.max(1).values.argmax()), float((lq - lqo).abs().max()), float(lqo.abs().max())))
a, b = res[(2, 4)], res[(1, 4)]
print("fused vs stream: x %.3e  log q %.3e" % (float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max())))
