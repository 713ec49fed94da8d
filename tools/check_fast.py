"""Fast mode against its CPU emulation: the oracle flow in float64 with the W x W Linear of every conditioner fed
bf16-rounded weights and bf16-rounded inputs (what the kernels' bf16 GEMMs compute, up to fp32 accumulation order)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa
from oracle import flow as oflow
from bench import build_flow_state

DEV = "cuda"


class Bf16Linear(torch.nn.Module):
    def __init__(self, lin):
        super().__init__()
        self.w = lin.weight.detach().float().bfloat16().double()
        self.b = lin.bias.detach().double()

    def forward(self, x):
        return x.float().bfloat16().double() @ self.w.t() + self.b


def emulate(nf):
    nf64 = copy.deepcopy(nf).double()
    for f in nf64.flows:
        if isinstance(f, oflow.AffineCouplingBlock):
            net = f.flows[1].param_map.net
            net[2] = Bf16Linear(net[2])
    return nf64


if __name__ == "__main__":
    D, K, nodes, B = [int(v) for v in os.environ.get("CFG", "32,10,10,256").split(",")]
    if (D, K, nodes) == (32, 10, 10):
        hf = build_flow_state(0)
    else:
        torch.manual_seed(0)
        hf = fa.RealNVP(D, K, nodes)
        with torch.no_grad():
            for l1, l2, l3, aff in hf._layers():
                l3.weight.normal_(0, 0.02); l3.bias.normal_(0, 0.02)
    nf = oflow.make_realnvp(D, K, nodes)
    nf.load_state_dict(hf._nf_model.state_dict())
    hf = hf.to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    torch.manual_seed(1)
    x, _ = hf.sample_and_log_prob((B,))
    p32 = fa.create_point(x, hf, target, with_grad=True)
    outs = []
    with fa.fast_mode():
        for _ in range(3):
            outs.append(fa.create_point(x, hf, target, with_grad=True))
    print("deterministic:", all(torch.equal(outs[0].log_q, o.log_q) and torch.equal(outs[0].grad_log_q, o.grad_log_q) for o in outs))
    em = emulate(nf)
    xg = x.cpu().double().requires_grad_(True)
    lq_e = em.log_prob(xg)
    (g_e,) = torch.autograd.grad(lq_e.sum(), xg)
    lq64 = copy.deepcopy(nf).double().log_prob(x.cpu().double()).detach()
    f = outs[0]
    print("fp32 HIP  vs float64 oracle : max |d log q| %.3e" % float((p32.log_q.cpu().double() - lq64).abs().max()))
    print("fast HIP  vs float64 oracle : max |d log q| %.3e" % float((f.log_q.cpu().double() - lq64).abs().max()))
    print("emulation vs float64 oracle : max |d log q| %.3e" % float((lq_e.detach() - lq64).abs().max()))
    print("fast HIP  vs emulation      : max |d log q| %.3e" % float((f.log_q.cpu().double() - lq_e.detach()).abs().max()))
    gn = g_e.norm(dim=1)
    print("fast HIP  vs emulation grad : max rel L2 %.3e" % float(((f.grad_log_q.cpu().double() - g_e).norm(dim=1) / gn).max()))
