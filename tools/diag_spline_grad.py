"""Diagnostic: d log q / dx of the 60-D / 12-layer spline flow - HIP and the fp32 CPU oracle against the float64 oracle."""
import copy, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_spline import make_pair, CASES, DEV

for D, L, hidden, circ, B in CASES[-1:]:
    of, hf = make_pair(D, L, hidden, circ, seed=D + L)
    g = torch.Generator().manual_seed(5)
    u, eps = torch.rand(B, D, generator=g), torch.randn(B, D, generator=g)
    with torch.no_grad():
        x_o, _ = of.sample_eps(u, eps)
    x = x_o + 0.3 * torch.randn(B, D, generator=g)
    x[0] = 7.0
    if len(circ):
        x[1, list(circ)] += 4 * math.pi
    of64 = copy.deepcopy(of).double()
    def grad(f, xx):
        xx = xx.clone().requires_grad_(True)
        lq = f.log_prob(xx)
        return lq.detach(), torch.autograd.grad(lq.sum(), xx)[0]
    lq32, g32 = grad(of, x)
    lq64, g64 = grad(of64, x.double())
    lqh, gh = hf.log_prob_and_grad(x.to(DEV))
    gh = gh.cpu().double()
    n64 = g64.norm(dim=1)
    rh = (gh - g64).norm(dim=1) / n64
    r32 = (g32.double() - g64).norm(dim=1) / n64
    print("grad norm range", float(n64.min()), float(n64.max()))
    print("rel L2 HIP vs f64   :", " ".join(f"{v:.1e}" for v in rh.tolist()))
    print("rel L2 o32 vs f64   :", " ".join(f"{v:.1e}" for v in r32.tolist()))
    print("log q err HIP/o32 vs f64:", float((lqh.cpu().double() - lq64).abs().max()), float((lq32.double() - lq64).abs().max()))
    # per-layer ablation: which coordinates carry the error for the worst sample
    b = int(rh.argmax())
    print("worst sample", b, "abs err per coord:", " ".join(f"{v:.1e}" for v in (gh[b] - g64[b]).abs().tolist()))
    print("   g64:", " ".join(f"{v:.2e}" for v in g64[b].tolist()))
