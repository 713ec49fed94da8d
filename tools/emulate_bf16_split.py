"""Lever (c) of VERDICT r1 #5, priced on the CPU before any kernel work: what a split-bf16 W x W GEMM
(hi = bf16(x), lo = bf16(x - hi); a.b ~ hi.hi + hi.lo + lo.hi, fp32 accumulate) does to log q and d log q / dx of the
headline RealNVP (10 x 16-320-320-32).  Output recorded in DESIGN.md section 4."""
import copy, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow as oflow

torch.manual_seed(0)
D, K, nodes, B = 32, 10, 10, 256
nf = oflow.make_realnvp(D, K, nodes); oflow.randomize_last_layers(nf, 0.05, 1)
x = nf.sample_eps(torch.randn(B, D))[0].detach() + 0.1 * torch.randn(B, D)


def split(t):
    hi = t.to(torch.bfloat16).float()
    return hi, (t - hi).to(torch.bfloat16).float()


class SplitLinear(torch.nn.Module):
    def __init__(self, lin):
        super().__init__(); self.lin = lin

    def forward(self, a):
        ah, al = split(a); wh, wl = split(self.lin.weight.t())
        return ah @ wh + ah @ wl + al @ wh + self.lin.bias


def variant(which):
    m = copy.deepcopy(nf)
    for k in range(K):
        net = m.flows[2 * k].flows[1].param_map.net
        for idx in which:
            net[idx] = SplitLinear(net[idx])
    return m


def lq_g(m, x):
    xg = x.clone().requires_grad_(True); l = m.log_prob(xg); g, = torch.autograd.grad(l.sum(), xg); return l.detach(), g


l64, g64 = lq_g(copy.deepcopy(nf).double(), x.double())
for name, m in (("fp32", nf), ("bf16 2-term split, W2 only", variant([2])), ("bf16 2-term split, all three", variant([0, 2, 4]))):
    l, g = lq_g(m, x)
    el = ((l.double() - l64).abs() / l64.abs().clamp(min=1)).max().item()
    eg = ((g.double() - g64).abs().max() / g64.abs().max()).item()
    print(f"{name:32s} log q rel err {el:.2e}   grad err / max|grad| {eg:.2e}")
