"""RealNVP with ActNorm layers (`make_wrapped_normflow_realnvp(act_norm=True)`, the reference builder's default,
experiments/make_flow/make_normflow_model.py:27-29,82-96) on the GPU: the kernels see every ActNorm folded into the
InvertibleAffine before it (csrc/flow_kernels.hip k_affine_assemble), its gradients come from the LU chain-rule kernel
(csrc/train_kernels.hip k_affine_grads).  Oracle: oracle/flow.py ActNorm (normflows' published definition; unpinned)."""
import copy

import pytest
import torch

from helpers import close, worst, RTOL
from test_gpu_parity import hip_relu_decisions, _ForcedReLU, _param_grads

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from oracle import ais as oais            # noqa: E402
from oracle import flow as oflow          # noqa: E402
from oracle import targets as otgt        # noqa: E402

DEV = "cuda"


def seeded_actnorm_flow(D, K, nodes, seed, std=0.05):
    torch.manual_seed(seed)
    nf = oflow.make_realnvp(D, K, nodes, act_norm=True)
    oflow.randomize_last_layers(nf, std, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    with torch.no_grad():
        for f in nf.flows:
            if isinstance(f, oflow.ActNorm):
                f.s.copy_(0.3 / K ** 0.5 * torch.randn(1, D, generator=g))     # (deep flows stay well-conditioned)
                f.t.copy_(0.5 / K ** 0.5 * torch.randn(1, D, generator=g))
                f.data_dep_init_done.fill_(1.0)
    return nf


def hip_from_oracle(nf):
    D = nf.q0.loc.shape[1]
    K = len(nf.flows) // 3
    W = nf.flows[0].flows[1].param_map.net[0].weight.shape[0]
    f = fa.RealNVP(D, K, W // D, act_norm=True)
    res = f._nf_model.load_state_dict(nf.state_dict(), strict=True)          # same keys as the normflows modules
    assert not res.missing_keys and not res.unexpected_keys
    return f.to(DEV).requires_grad_(False)


@pytest.mark.parametrize("D,K,nodes,B", [(6, 3, 5, 50), (32, 10, 10, 48), (2, 4, 40, 100), (60, 2, 4, 20), (5, 2, 4, 17),
                                         (64, 2, 8, 24)])
def test_actnorm_flow_log_prob_grad_sample_vs_oracle(D, K, nodes, B):
    nf = seeded_actnorm_flow(D, K, nodes, 40 + D + K)
    hf = hip_from_oracle(nf)
    torch.manual_seed(5)
    eps = torch.randn(B, D)
    with torch.no_grad():
        x_o, lq_s_o = nf.sample_eps(eps)
    x_h, lq_s_h = hf.native_sample(eps.to(DEV))
    assert close(x_h, x_o, RTOL), f"sample x: {worst(x_h, x_o):.2f}x tol"
    assert close(lq_s_h, lq_s_o, RTOL), f"sample log q: {worst(lq_s_h, lq_s_o):.2f}x tol"
    x = x_o + 0.1 * torch.randn(B, D)
    xg = x.clone().requires_grad_(True)
    lq_o = nf.log_prob(xg)
    (g_o,) = torch.autograd.grad(lq_o.sum(), xg)
    lq_h, g_h = hf.log_prob_and_grad(x.to(DEV))
    assert close(lq_h, lq_o.detach(), RTOL), f"log q: {worst(lq_h, lq_o.detach()):.2f}x tol"
    assert close(g_h, g_o, RTOL, atol_scale=10), f"grad: {worst(g_h, g_o):.2f}x tol"
    assert close(hf.log_prob(x_h), lq_s_h, RTOL)
    # an ActNorm with s = t = 0 is the identity: same numbers as the flow without ActNorm layers
    plain = oflow.make_realnvp(D, K, nodes)
    sd = {k: v for k, v in nf.state_dict().items()}
    remap = {}
    for k, v in sd.items():
        i = int(k.split(".")[1]) if k.startswith("flows.") else None
        if i is None:
            remap[k] = v
        elif i % 3 != 2:
            remap[".".join(["flows", str(2 * (i // 3) + i % 3)] + k.split(".")[2:])] = v
    plain.load_state_dict(remap)
    hp = fa.RealNVP(D, K, nodes)
    hp._nf_model.load_state_dict(plain.state_dict())
    hp = hp.to(DEV)
    with torch.no_grad():
        for an in hf._act_norms():
            an.s.zero_(); an.t.zero_()
    a, b = hf.log_prob_and_grad(x.to(DEV)), hp.log_prob_and_grad(x.to(DEV))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_data_dependent_initialisation_matches_the_first_sample_call_of_the_oracle():
    """make_wrapped_normflow_realnvp(act_norm=True) -> `wrapped_dist.sample((500,))` (make_normflow_model.py:94-95):
    each ActNorm takes s = -log(std + 1e-6), t = -mean * exp(s) from the batch that reaches it."""
    D, K, nodes, N = 6, 4, 5, 500
    torch.manual_seed(1)
    nf = oflow.make_realnvp(D, K, nodes, act_norm=True)
    oflow.randomize_last_layers(nf, 0.2, 3)
    hf = hip_from_oracle(nf)                                   # before the oracle's ActNorms are initialised
    assert all(float(an.data_dep_init_done) == 0 for an in hf._act_norms())
    eps = torch.randn(N, D)
    with torch.no_grad():
        x_o, lq_o = nf.sample_eps(eps)                         # initialises the oracle's ActNorm layers on the way
    hf.init_act_norm(eps=eps.to(DEV))
    for k, an in enumerate(hf._act_norms()):
        ref = nf.flows[3 * k + 2]
        assert float(an.data_dep_init_done) == 1.0
        assert close(an.s.detach(), ref.s.detach(), RTOL), f"layer {k} s: {worst(an.s.detach(), ref.s.detach()):.2f}x"
        assert close(an.t.detach(), ref.t.detach(), RTOL, atol_scale=10), f"layer {k} t"
    x_h, lq_h = hf.native_sample(eps.to(DEV))
    assert close(x_h, x_o, RTOL, atol_scale=10) and close(lq_h, lq_o, RTOL)
    # per-coordinate statistics of the initialising batch after the last ActNorm: zero mean, unit std
    assert float(x_h.mean(0).abs().max()) < 1e-4 and float((x_h.std(0) - 1).abs().max()) < 1e-4
    # the builder initialises lazily, the first time the flow is used on the GPU
    torch.manual_seed(0)
    lazy = fa.make_wrapped_normflow_realnvp(D, K, nodes).to(DEV)           # act_norm=True is the reference's default
    assert lazy.act_norm and "_nf_model.flows.2.s" in lazy.state_dict()
    x, lq = lazy.sample_and_log_prob((256,))
    assert all(float(an.data_dep_init_done) == 1.0 for an in lazy._act_norms()) and torch.isfinite(lq).all()
    assert close(lazy.log_prob(x), lq, RTOL)


@pytest.mark.parametrize("D,K,nodes,B", [(6, 3, 5, 50), (32, 10, 10, 333), (2, 4, 40, 100), (60, 2, 4, 40)])
def test_actnorm_flow_parameter_gradients_vs_oracle_autograd(D, K, nodes, B):
    """sum_b c_b d log q(x_b) / d theta for every parameter INCLUDING ActNorm's s and t, against autograd through a
    float64 copy of the oracle whose ReLUs take the decisions of the HIP forward (as in test_gpu_parity)."""
    nf = seeded_actnorm_flow(D, K, nodes, 300 + D + K)
    hf = hip_from_oracle(nf).requires_grad_(True)
    torch.manual_seed(11)
    with torch.no_grad():
        x = nf.sample_eps(torch.randn(B, D))[0] + 0.1 * torch.randn(B, D)
    coef = torch.randn(B) / B
    nf64 = copy.deepcopy(nf).double()
    for k, (m1, m2) in enumerate(hip_relu_decisions(hf, x.to(DEV))):
        net = nf64.flows[3 * k].flows[1].param_map.net
        net[1], net[3] = _ForcedReLU(m1), _ForcedReLU(m2)
    names = [n for n, _ in nf.named_parameters()]
    lq_o, g_o, gx_o = _param_grads(nf64, [p for _, p in nf64.named_parameters()], x.double(), coef.double(), x_grad=True)
    hip_params = dict(hf._nf_model.named_parameters())
    assert set(hip_params) == set(names)
    plist = [hip_params[n] for n in names]
    lq_h, g_h, gx_h = _param_grads(hf, plist, x.to(DEV), coef.to(DEV), x_grad=True)
    assert close(lq_h, lq_o.float(), RTOL) and close(gx_h, gx_o.float(), RTOL, atol_scale=10)
    for n, a, b in zip(names, g_h, g_o):
        assert a.shape == b.shape
        assert close(a, b.float(), RTOL, atol_scale=30), f"{n}: {worst(a, b.float()):.2f} x tolerance vs the fp64 oracle"
    assert any(n.endswith(".s") for n in names) and any(n.endswith(".t") for n in names)
    # FlatAdam: the flat image carries the ActNorm pairs behind the base block
    opt = fa.FlatAdam(hf, lr=1e-3)
    assert opt.theta.numel() == sum(p.numel() for p in hf.parameters()) == hf.grad_floats()
    before = opt.theta.detach().clone()
    opt.zero_grad()
    (hf.log_prob(x.to(DEV)) * coef.to(DEV)).sum().backward()
    flat = opt.theta.grad.detach().clone()
    for n, v in zip([n for n, _ in hf._nf_model.named_parameters()], hf._grad_views(flat)):
        pass
    views = hf._grad_views(flat)
    by_id = {id(p): v for p, v in zip(hf._grad_tensors(), views)}
    for n, b in zip(names, g_o):
        assert close(by_id[id(hip_params[n])], b.float(), RTOL, atol_scale=30), n
    opt.step(max_grad_norm=5.0)
    assert not torch.equal(opt.theta.detach(), before)


def test_hmc_transitions_and_fused_ais_with_an_actnorm_flow_vs_oracle():
    """The fused transition / AIS kernels evaluate the folded flow too: teacher-forced per-transition parity and the
    whole-chain call against the oracle sampler (as test_ais_headline_config_vs_oracle)."""
    D, K, M, B = 32, 10, 4, 32
    nf = seeded_actnorm_flow(D, K, 10, 9, std=0.03)
    with torch.no_grad():
        for f in nf.flows:
            if isinstance(f, oflow.ActNorm):
                f.s.mul_(0.2); f.t.mul_(0.2)
    hf = hip_from_oracle(nf)
    target, otarget = fa.ManyWellEnergy(D), otgt.ManyWell(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1,
                                   eval_mode=True).to(DEV)
    assert hmc.is_native
    ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
    torch.manual_seed(3)
    eps0 = torch.randn(B, D); noise_p = torch.randn(M, 1, B, D); noise_e = torch.empty(M, 1, B).exponential_()
    ohmc = oais.HMC(M, D, nf.log_prob, otarget.log_prob, alpha=2.0, p_target=False, epsilon=0.1, eval_mode=True)
    oa = oais.AIS(lambda e: tuple(t.detach() for t in nf.sample_eps(e)), nf.log_prob, otarget.log_prob, ohmc, False, 2.0, M)
    opt, olw, _ = oa.sample_and_log_weights(eps0, noise_p, noise_e, keep_snapshots=True)
    snaps = oa.snapshots
    for j in range(1, M + 1):
        p_in, lw_in = snaps[j - 1]
        p_ref, lw_ref = snaps[j]
        pt = fa.Point(p_in.x.clone().to(DEV), p_in.log_q.clone().to(DEV), p_in.log_p.clone().to(DEV),
                      p_in.grad_log_q.clone().to(DEV), p_in.grad_log_p.clone().to(DEV))
        lw = lw_in.clone().to(DEV)
        hmc.transition(pt, j, float(ais.B_space[j]), log_w=lw, beta_next=float(ais.B_space[j + 1]),
                       noise_p=noise_p[j - 1].to(DEV), noise_e=noise_e[j - 1].to(DEV))
        scale = max(1.0, float(p_ref.x.abs().max()))
        err = (pt.x.cpu() - p_ref.x).abs().max(1).values / scale
        ok = err <= 1e-4
        assert (~ok).sum() <= 1, f"transition {j}: {int((~ok).sum())} chains differ (max err {float(err.max()):.2e})"
        assert close(lw.cpu()[ok], lw_ref[ok], RTOL) and close(pt.log_q.cpu()[ok], p_ref.log_q[ok], RTOL)
    pt, log_w = ais.sample_and_log_weights(B, eps0=eps0.to(DEV), noise_a=noise_p.to(DEV), noise_b=noise_e.to(DEV))
    same = (pt.x.cpu() - opt.x).abs().max(1).values < 1e-2
    assert same.sum() >= B - 4 and close(log_w.cpu()[same], olw[same], 1e-3)
