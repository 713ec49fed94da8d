"""Backward of the SAMPLING direction (`loss.backward()` through `flow.sample_and_log_prob`, the reparameterised baseline
losses flow_reverse_kl / flow_alpha_2_div_nis of fab/core.py:130-152): fabhip::realnvp_sample_tape (HIP sampler forward,
k_flow_sample_bwd + the parameter-gradient kernels backward) against float64 autograd through the oracle flow on the same
base noise.  Tolerance: per-tensor relative L2 error <= 2e-3 (fp32 kernels vs float64; a ReLU decision that differs
between the two for one sample moves a tensor's gradient by ~1/B of its norm)."""
import copy

import pytest
import torch

import fab_torch_amd as fa
from oracle import flow as oflow

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
RTOL = 2e-3


def make_pair(D, K, nodes, act_norm, seed):
    torch.manual_seed(seed)
    nf = oflow.make_realnvp(D, K, nodes, act_norm=act_norm)
    oflow.randomize_last_layers(nf, 0.05 / max(1, K) ** 0.5, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    with torch.no_grad():
        nf.q0.loc.copy_(0.3 * torch.randn(1, D, generator=g))
        nf.q0.log_scale.copy_(0.2 * torch.randn(1, D, generator=g))
        for f in nf.flows:
            if isinstance(f, oflow.ActNorm):
                f.s.copy_(0.2 * torch.randn(1, D, generator=g) / K ** 0.5)
                f.t.copy_(0.3 * torch.randn(1, D, generator=g) / K ** 0.5)
                f.data_dep_init_done.fill_(1.0)
    hf = fa.RealNVP(D, K, nodes, act_norm=act_norm)
    hf._nf_model.load_state_dict(nf.state_dict(), strict=True)
    return copy.deepcopy(nf).double(), hf.to(DEV)


def loss_of(x, lq, a, c, w):
    return (a * lq).sum() + (c * x).sum() + 0.5 * (w[:, None] * x * x).sum()


@pytest.mark.parametrize("D,K,nodes,act_norm,B", [
    (6, 2, 6, False, 64), (5, 3, 8, False, 37), (32, 4, 10, False, 96), (32, 3, 10, True, 64),
    (60, 3, 4, False, 48), (2, 2, 40, True, 33), (16, 2, 32, False, 50)])
def test_sample_direction_parameter_gradients_match_float64_autograd(D, K, nodes, act_norm, B):
    nf64, hf = make_pair(D, K, nodes, act_norm, seed=11 + D + K)
    g = torch.Generator().manual_seed(5)
    eps = torch.randn(B, D, generator=g)
    a, c, w = torch.randn(B, generator=g), torch.randn(B, D, generator=g), torch.rand(B, generator=g)
    # float64 reference
    e64 = eps.double().requires_grad_(True)
    x64, lq64 = nf64.sample_eps(e64)
    loss_of(x64, lq64, a.double(), c.double(), w.double()).backward()
    ref = {n: p.grad for n, p in nf64.named_parameters()}
    # HIP
    ed = eps.to(DEV).requires_grad_(True)
    x, lq = hf.sample_and_log_prob((B,), eps=ed)
    assert x.requires_grad and lq.requires_grad
    assert float((x.detach().cpu().double() - x64.detach()).abs().max().item()) <= 1e-4 * max(1.0, float(x64.detach().abs().max()))
    assert float((lq.detach().cpu().double() - lq64.detach()).abs().max()) <= 1e-4 * max(1.0, float(lq64.detach().abs().max()))
    loss_of(x, lq, a.to(DEV), c.to(DEV), w.to(DEV)).backward()
    worst = 0.0
    for n, p in hf._nf_model.named_parameters():
        r = ref[n]
        assert p.grad is not None, n
        num = float((p.grad.cpu().double() - r).norm())
        den = float(r.norm())
        if den < 1e-9:
            assert num < 1e-6, n
            continue
        worst = max(worst, num / den)
        assert num / den <= RTOL, (n, num / den)
    ge = ed.grad.cpu().double()
    assert float((ge - e64.grad).norm() / e64.grad.norm()) <= RTOL
    assert worst > 0.0


def test_reverse_kl_gradient_matches_float64_autograd():
    """fab/core.py:130-133: loss = mean(log q(x) - log p(x)), x ~ q reparameterised; ManyWell-32 headline flow shape."""
    D, K, nodes, B = 32, 4, 10, 128
    nf64, hf = make_pair(D, K, nodes, False, seed=3)
    target = fa.ManyWellEnergy(D)
    from oracle import targets as otargets
    tgt64 = otargets.ManyWell(D)
    eps = torch.randn(B, D, generator=torch.Generator().manual_seed(9))
    x64, lq64 = nf64.sample_eps(eps.double())
    (lq64 - tgt64.log_prob(x64)).mean().backward()
    x, lq = hf.sample_and_log_prob((B,), eps=eps.to(DEV))
    (lq - target.log_prob(x)).mean().backward()
    for (n, p), (_, r) in zip(hf._nf_model.named_parameters(), nf64.named_parameters()):
        den = float(r.grad.norm())
        if den > 1e-9:
            assert float((p.grad.cpu().double() - r.grad).norm()) / den <= RTOL, n


def test_no_grad_sampling_is_unchanged_and_bitwise_equal_to_the_differentiable_forward():
    nf64, hf = make_pair(32, 3, 10, False, seed=2)
    eps = torch.randn(80, 32, device=DEV)
    with torch.no_grad():
        x0, l0 = hf.sample_and_log_prob((80,), eps=eps)
    x1, l1 = hf.sample_and_log_prob((80,), eps=eps)
    assert torch.equal(x0, x1.detach()) and torch.equal(l0, l1.detach())
