"""Fast mode (fabhip_set_fast_mode / fab_torch_amd.fast_mode): the W x W GEMMs of the RealNVP conditioners on the bf16
matrix cores inside the transition kernels.  NOT the parity path - these tests pin what it IS: the flow with bf16-rounded
W2 weights and bf16-rounded inputs of that Linear (fp32 accumulation), deterministic, off by default, leaving the fp32
path bit-for-bit alone, and a sampler whose ESS on the trained-flow fixture stays within 1 % of the fp32 path's."""
import copy

import numpy as np
import pytest
import torch

from helpers import load_golden, close, RTOL

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from oracle import flow as oflow          # noqa: E402

DEV = "cuda"


class _Bf16Linear(torch.nn.Module):
    def __init__(self, lin):
        super().__init__()
        self.w = lin.weight.detach().float().bfloat16().double()
        self.b = lin.bias.detach().double()

    def forward(self, x):
        return x.float().bfloat16().double() @ self.w.t() + self.b


def emulation(nf):
    nf64 = copy.deepcopy(nf).double()
    for f in nf64.flows:
        if isinstance(f, oflow.AffineCouplingBlock):
            net = f.flows[1].param_map.net
            net[2] = _Bf16Linear(net[2])
    return nf64


@pytest.mark.parametrize("D,K,nodes,B", [(32, 10, 10, 128), (6, 8, 40, 64), (2, 4, 40, 100), (32, 2, 16, 40), (64, 2, 8, 24),
                                         (6, 3, 5, 50)])
def test_fast_mode_is_the_flow_with_bf16_rounded_inner_gemm_operands(D, K, nodes, B):
    torch.manual_seed(D + K)
    nf = oflow.make_realnvp(D, K, nodes)
    oflow.randomize_last_layers(nf, 0.02, 5)
    hf = fa.RealNVP(D, K, nodes)
    hf._nf_model.load_state_dict(nf.state_dict())
    hf = hf.to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    x, _ = hf.sample_and_log_prob((B,))
    assert int(fa._ops.load().get_fast_mode()) == 0                      # off by default
    p32 = fa.create_point(x, hf, target, with_grad=True)
    with fa.fast_mode():
        assert int(fa._ops.load().get_fast_mode()) == 1
        pf = fa.create_point(x, hf, target, with_grad=True)
        pf2 = fa.create_point(x, hf, target, with_grad=True)
        lq_f, g_f = hf.log_prob_and_grad(x)                                # the gradient-returning density op follows the mode
        lq_plain = hf.log_prob(x)                                          # a plain density evaluation stays fp32
    assert int(fa._ops.load().get_fast_mode()) == 0
    p32b = fa.create_point(x, hf, target, with_grad=True)
    assert torch.equal(p32.log_q, p32b.log_q) and torch.equal(p32.grad_log_q, p32b.grad_log_q)   # fp32 path untouched
    assert torch.equal(pf.log_q, pf2.log_q) and torch.equal(pf.grad_log_q, pf2.grad_log_q)       # deterministic
    assert torch.equal(lq_f, pf.log_q) and torch.equal(g_f, pf.grad_log_q)
    assert close(lq_plain, p32.log_q, RTOL)
    assert torch.equal(pf.log_p, p32.log_p) and torch.equal(pf.grad_log_p, p32.grad_log_p)       # the target is not touched
    em = emulation(nf)
    xg = x.cpu().double().requires_grad_(True)
    lq_e = em.log_prob(xg)
    (g_e,) = torch.autograd.grad(lq_e.sum(), xg)
    lq_e = lq_e.detach()
    dev_em = float((pf.log_q.cpu().double() - lq_e).abs().max())
    dev_32 = float((pf.log_q.cpu().double() - p32.log_q.cpu().double()).abs().max())
    # the kernels against the emulation: fp32 accumulation order + activations within rounding of a bf16 tie; against
    # the fp32 path: the bf16 approximation itself (a few 1e-3 of |log q| ~ 10 .. 50)
    assert dev_em <= 2e-3, f"fast mode vs its emulation: {dev_em:.2e}"
    assert 0 < dev_32 <= 5e-2, f"fast mode vs fp32: {dev_32:.2e}"
    rel = (pf.grad_log_q.cpu().double() - g_e).norm(dim=1) / g_e.norm(dim=1)
    assert float(rel.median()) <= 2e-3 and float(rel.max()) <= 5e-2, f"grad vs emulation: {float(rel.max()):.2e}"


@pytest.mark.parametrize("D,K,nodes,B,M", [(32, 10, 10, 1024, 2), (32, 10, 10, 37, 3), (6, 8, 40, 64, 2), (16, 3, 8, 100, 2)])
def test_fast_mode_on_four_chain_tiles_is_the_same_flow_with_bf16_rounded_inner_gemm_operands(D, K, nodes, B, M):
    """Round 5: up to 1152 chains fast mode runs on the 4-chain tiles with fused stages (flow_r4f.h, bf16 W x W tiles,
    v_mfma_f32_4x4x4_16b_bf16) - chain initialisation and transitions of a fused AIS call.  log q and d log q / dx of the
    returned points (written by those kernels) against the float64 emulation at the returned x; deterministic; the fp32
    call on the same noise untouched by the switch."""
    torch.manual_seed(D + K)
    nf = oflow.make_realnvp(D, K, nodes)
    oflow.randomize_last_layers(nf, 0.02, 5)
    hf = fa.RealNVP(D, K, nodes)
    hf._nf_model.load_state_dict(nf.state_dict())
    hf = hf.to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(7)
    eps0 = torch.randn(B, D, device=DEV, generator=g)
    na = torch.randn(M, 1, B, D, device=DEV, generator=g)
    nb = torch.empty(M, 1, B, device=DEV).exponential_(generator=g)

    def call(fast):
        hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=3,
                                       eval_mode=True).to(DEV)
        ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
        with fa.fast_mode(fast):
            pt, lw = ais.sample_and_log_weights(B, eps0=eps0, noise_a=na, noise_b=nb)
        return pt.x.clone(), pt.log_q.clone(), pt.grad_log_q.clone(), lw.clone()

    r32, rf, rf2, r32b = call(False), call(True), call(True), call(False)
    assert all(torch.equal(a, b) for a, b in zip(r32, r32b))              # the fp32 path is untouched
    assert all(torch.equal(a, b) for a, b in zip(rf, rf2))                # deterministic
    assert not torch.equal(rf[1], r32[1])                                 # (and it is another computation)
    x, lq, gq, _ = rf
    em = emulation(nf)
    xg = x.cpu().double().requires_grad_(True)
    lq_e = em.log_prob(xg)
    (g_e,) = torch.autograd.grad(lq_e.sum(), xg)
    dev_em = float((lq.cpu().double() - lq_e.detach()).abs().max())
    assert dev_em <= 2e-3, f"4-chain fast mode vs its emulation: {dev_em:.2e}"
    rel = (gq.cpu().double() - g_e).norm(dim=1) / g_e.norm(dim=1)
    assert float(rel.median()) <= 2e-3 and float(rel.max()) <= 5e-2, f"grad vs emulation: {float(rel.max()):.2e}"


def test_fast_mode_ais_on_the_trained_flow_keeps_the_ess_within_one_percent():
    """g13: the committed trained ManyWell-6 flow and the reference's evaluation AIS call on it (captured noise).  The
    fp32 path reproduces the reference ESS to 1e-6 (test_gpu_workloads); the fast mode must stay within 1 % of it."""
    g = load_golden("g13_trained_flow_mw6.npz")
    D, K, nodes, M, L, B = int(g["D"]), int(g["K"]), int(g["nodes"]), int(g["M"]), int(g["L"]), int(g["B"])
    flow = fa.RealNVP(D, K, nodes)
    flow._nf_model.load_state_dict({k[len("flow."):]: torch.tensor(v) for k, v in g.items() if k.startswith("flow.")})
    flow = flow.to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    for tag, p_target in (("p", True), ("g", False)):
        res = {}
        for fast in (False, True):
            hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=float(g["alpha"]), p_target=p_target,
                                           epsilon=1.0, L=L, eval_mode=True).to(DEV)
            with torch.no_grad():
                hmc.epsilons.copy_(torch.tensor(g["epsilons"])); hmc.common_epsilon.copy_(torch.tensor(g["common_epsilon"]))
            ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target, float(g["alpha"]), M)
            T = lambda k: torch.tensor(g[k]).to(DEV)      # noqa: E731
            with fa.fast_mode(fast):
                pt, lw = ais.sample_and_log_weights(B, eps0=T(f"{tag}_eps0"), noise_a=T(f"{tag}_noise_p"),
                                                    noise_b=T(f"{tag}_noise_e"))
            res[fast] = (ais.get_logging_info(), pt.x.clone(), lw.clone())
        ref = float(g[f"{tag}_ess_ais"])
        assert abs(res[False][0]["ess_ais"] - ref) / ref < 1e-3
        assert not torch.equal(res[True][2], res[False][2]) and torch.isfinite(res[True][2]).all()
        if p_target:
            rel = abs(res[True][0]["ess_ais"] - ref) / ref
            assert rel < 1e-2, f"target p: fast-mode ESS {res[True][0]['ess_ais']:.5f} vs reference {ref:.5f}"
            assert abs(res[True][0]["log_Z"] - res[False][0]["log_Z"]) < 2e-2
        else:
            # target p^2/q: the reference's own ESS is 0.02 (a handful of chains carry the weight) and HMC at the tuned step
            # size is chaotic - half of the chains leave the fp32 trajectory after 8 x 5 leapfrogs under a 1e-3
            # perturbation of log q - so this ESS is a different draw of a heavy-tailed statistic, not a parity quantity
            assert 0 < res[True][0]["ess_ais"] <= 1 and abs(res[True][0]["log_Z"] - res[False][0]["log_Z"]) < 1.0


@pytest.mark.parametrize("D,L,hidden,circ,B", [(8, 4, 64, (1, 4, 6), 64), (32, 6, 256, (), 48), (60, 4, 256, (3, 7, 20, 41), 32),
                                               (7, 3, 128, (0, 6), 33)])
def test_fast_mode_of_the_spline_flow_is_the_conditioner_with_bf16_rounded_inner_gemm_operands(D, L, hidden, circ, B):
    """The spline family's fast mode: blocks.0.linear_layers.{0,1} and final_layer of every conditioner on the bf16 matrix
    cores in the one-launch density + gradient kernel (gradient evaluations only)."""
    import math
    from oracle import spline as osp
    tb = torch.full((D,), 5.0); tb[list(circ)] = math.pi
    torch.manual_seed(D + L)
    of = osp.make_circular_coupled_flow(D, L, hidden, circ, tb, seed=3)
    osp.randomize(of, 0.2, 4)
    hf = fa.CircularCoupledRQSFlow(D, L, hidden, circ, tb, seed=3)
    hf._nf_model.load_state_dict(of.state_dict(), strict=True)
    hf = hf.to(DEV).requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        x, _ = of.sample_eps(torch.rand(B, D, generator=g), torch.randn(B, D, generator=g))
    x = x + 0.2 * torch.randn(B, D, generator=g)
    from fab_torch_amd import _ops
    lq32, g32 = hf.log_prob_and_grad(x.to(DEV))
    if hidden > 192:
        # hidden widths padded to 256 have the fp32 stream kernel (spline_r8.h), which is FASTER than the bf16 kernel up to
        # 8 chains per CU: fast mode (a permission, not an order) takes it there - same bits as the parity path
        with fa.fast_mode():
            lq_s, g_s = hf.log_prob_and_grad(x.to(DEV))
        assert torch.equal(lq_s, lq32) and torch.equal(g_s, g32)
    # the bf16 kernel itself (the 16x16x4 family; what fast mode runs for other widths and above 8 chains per CU)
    with _ops.option(_ops.OPT_SPLINE_MFMA, 16):
        lq32, g32 = hf.log_prob_and_grad(x.to(DEV))
        with fa.fast_mode():
            lqf, gf = hf.log_prob_and_grad(x.to(DEV))
            lqf2, gf2 = hf.log_prob_and_grad(x.to(DEV))
            lq_plain = hf.log_prob(x.to(DEV))
        lq32b, g32b = hf.log_prob_and_grad(x.to(DEV))
    assert torch.equal(lq32, lq32b) and torch.equal(g32, g32b) and torch.equal(lqf, lqf2) and torch.equal(gf, gf2)
    assert close(lq_plain, lq32, 1e-5)
    em = copy.deepcopy(of).double()
    for f in em.flows:
        if isinstance(f, osp.CircularCoupledRationalQuadraticSpline):
            net = f.prqct.transform_net
            net.blocks[0].linear_layers[0] = _Bf16Linear(net.blocks[0].linear_layers[0])
            net.blocks[0].linear_layers[1] = _Bf16Linear(net.blocks[0].linear_layers[1])
            net.final_layer = _Bf16Linear(net.final_layer)
    xg = x.double().requires_grad_(True)
    lq_e = em.log_prob(xg)
    (g_e,) = torch.autograd.grad(lq_e.sum(), xg)
    lq_e = lq_e.detach()
    dev_em = float((lqf.cpu().double() - lq_e).abs().max())
    dev_32 = float((lqf.cpu().double() - lq32.cpu().double()).abs().max())
    assert dev_em <= 5e-3, f"fast mode vs its emulation: {dev_em:.2e}"
    assert 0 < dev_32 <= 0.2, f"fast mode vs fp32: {dev_32:.2e}"
    rel = (gf.cpu().double() - g_e).norm(dim=1) / g_e.norm(dim=1)
    assert float(rel.median()) <= 1e-2 and float(rel.max()) <= 0.2, f"grad vs emulation: {float(rel.median()):.2e} / {float(rel.max()):.2e}"


def test_fast_mode_with_actnorm_layers_matches_its_emulation():
    """The folded ActNorm terms (fp32 affine stages) and the bf16 W x W GEMMs together: fast mode of an act_norm=True flow
    against the float64 emulation, and a fused fast-mode AIS call on it."""
    D, K, nodes, B = 32, 4, 10, 64
    torch.manual_seed(3)
    nf = oflow.make_realnvp(D, K, nodes, act_norm=True)
    oflow.randomize_last_layers(nf, 0.02, 5)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for f in nf.flows:
            if isinstance(f, oflow.ActNorm):
                f.s.copy_(0.1 * torch.randn(1, D, generator=g)); f.t.copy_(0.2 * torch.randn(1, D, generator=g))
                f.data_dep_init_done.fill_(1.0)
    hf = fa.RealNVP(D, K, nodes, act_norm=True)
    hf._nf_model.load_state_dict(nf.state_dict(), strict=True)
    hf = hf.to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    x, _ = hf.sample_and_log_prob((B,))
    with fa.fast_mode():
        pf = fa.create_point(x, hf, target, with_grad=True)
    em = emulation(nf)
    xg = x.cpu().double().requires_grad_(True)
    lq_e = em.log_prob(xg)
    (g_e,) = torch.autograd.grad(lq_e.sum(), xg)
    assert float((pf.log_q.cpu().double() - lq_e.detach()).abs().max()) <= 2e-3
    rel = (pf.grad_log_q.cpu().double() - g_e).norm(dim=1) / g_e.norm(dim=1)
    assert float(rel.median()) <= 2e-3 and float(rel.max()) <= 5e-2
    hmc = fa.HamiltonianMonteCarlo(3, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=3).to(DEV)
    ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, 3)
    with fa.fast_mode():
        pt, lw = ais.sample_and_log_weights(256)
    assert pt.x.shape == (256, D) and torch.isfinite(lw).all()


def test_precision_is_a_per_call_choice_of_the_flow_not_only_a_process_switch():
    """`fabhip_set_fast_mode` alone is process-wide - two samplers of one process could not differ.
    `flow.precision` ("fp32" / "fast" / None = process default) travels with every call (fabhip_flow::precision): a "fast"
    flow and an "fp32" flow side by side, under either process default, each give bit for bit what the process switch gives."""
    D, K, nodes, M, B, L = 32, 4, 10, 3, 96, 3
    torch.manual_seed(3)
    nf = oflow.make_realnvp(D, K, nodes)
    oflow.randomize_last_layers(nf, 0.02, 5)
    flows = {}
    for prec in (None, "fp32", "fast"):
        hf = fa.RealNVP(D, K, nodes)
        hf._nf_model.load_state_dict(nf.state_dict())
        hf = hf.to(DEV).requires_grad_(False)
        hf.precision = prec
        flows[prec] = hf
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(1)
    eps0 = torch.randn(B, D, device=DEV, generator=g)
    na = torch.randn(M, 1, B, D, device=DEV, generator=g)
    nb = torch.empty(M, 1, B, device=DEV).exponential_(generator=g)

    def run(hf):
        hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=L).to(DEV)
        ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
        pt, lw = ais.sample_and_log_weights(B, eps0=eps0, noise_a=na, noise_b=nb)
        lq, gr = hf.log_prob_and_grad(pt.x)
        return pt.x.clone(), lw.clone(), lq.clone(), gr.clone(), hmc.epsilons.clone()

    ref32 = run(flows[None])                                    # process default off
    with fa.fast_mode():
        ref_fast = run(flows[None])                             # process default on
        in_fast_fp32 = run(flows["fp32"])                       # the flow's own choice wins over the switch
    own_fast = run(flows["fast"])                               # ... in both directions
    own_fp32 = run(flows["fp32"])
    for a, b in zip(ref32, own_fp32):
        assert torch.equal(a, b)
    for a, b in zip(ref32, in_fast_fp32):
        assert torch.equal(a, b)
    for a, b in zip(ref_fast, own_fast):
        assert torch.equal(a, b)
    assert not torch.equal(ref32[2], ref_fast[2])               # and the two modes do differ
    flows["fast"].precision = "bf16"
    with pytest.raises(fa._ops.FabhipError, match="precision"):
        flows["fast"].log_prob_and_grad(eps0)
