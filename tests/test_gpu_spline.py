"""RQ-spline coupling flow (R9s): HIP kernels (torch.ops.fabhip.spline_*) against oracle/spline.py on the same
parameters and noise - log q, d log q / dx, samples; the alanine-dipeptide shape (60-D, 12 layers, hidden 256, 8 bins,
12 circular coordinates); HMC transitions on a 60-D target through the generic plug-in path, per transition vs the
oracle.  (normflows is absent from the reference: the oracle is the specification, parity with the reference unpinned.)"""
import math

import numpy as np
import pytest
import torch

from helpers import close, worst, max_rel_err, RTOL

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from oracle import ais as oais            # noqa: E402
from oracle import spline as osp          # noqa: E402
from oracle import targets as otgt        # noqa: E402

DEV = "cuda"


def make_pair(D, L, hidden, circ, seed, std=0.4):
    tb = torch.full((D,), 5.0)
    g = torch.Generator().manual_seed(seed)
    tb[list(circ)] = math.pi / (0.5 + torch.rand(len(circ), generator=g))
    torch.manual_seed(seed)                                # nn.Linear's default init draws from the global generator
    of = osp.make_circular_coupled_flow(D, L, hidden, circ, tb, seed=seed)
    osp.randomize(of, std, seed + 1)
    hf = fa.CircularCoupledRQSFlow(D, L, hidden, circ, tb, seed=seed)
    missing = hf._nf_model.load_state_dict(of.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return of, hf.to(DEV).requires_grad_(False)


CASES = [(8, 4, 64, (1, 4, 6), 100), (6, 3, 32, (), 64), (7, 5, 128, (0, 6), 33),
         (60, 12, 256, (3, 7, 8, 12, 20, 21, 22, 30, 41, 45, 52, 59), 48),
         # hidden width padded to 256 = the 8-chain-tile kernel (spline_r8.h) with 2 / 1 / 4 chunks of conditioner outputs,
         # ragged batches (the last workgroup holds 5 / 3 / 1 chains), hidden < its padding
         (32, 6, 256, (), 77), (12, 3, 200, (2, 5), 19), (64, 2, 256, (0, 63), 9)]


@pytest.mark.parametrize("D,L,hidden,circ,B", CASES)
def test_spline_flow_log_prob_grad_and_sample_vs_oracle(D, L, hidden, circ, B):
    of, hf = make_pair(D, L, hidden, circ, seed=D + L)
    g = torch.Generator().manual_seed(5)
    u, eps = torch.rand(B, D, generator=g), torch.randn(B, D, generator=g)
    with torch.no_grad():
        x_o, lq_s_o = of.sample_eps(u, eps)
    x_h, lq_s_h = hf.sample_and_log_prob((B,), u=u.to(DEV), eps=eps.to(DEV))
    # the sampling direction INVERTS every spline (quadratic root 2c / (-b - sqrt(b^2 - 4ac)), ill-conditioned next to
    # a knot) through up to 12 layers: where fp32 itself is not good to 1e-4 the float64 oracle arbitrates - the HIP
    # worst HIP deviation from it may not exceed 1.5x the fp32 CPU oracle's own worst deviation
    import copy
    of64 = copy.deepcopy(of).double()
    with torch.no_grad():
        x64, lq64 = of64.sample_eps(u.double(), eps.double())
    for name, h, o32, o64 in (("x", x_h.cpu(), x_o, x64), ("log q", lq_s_h.cpu(), lq_s_o, lq64)):
        if close(h, o32, RTOL):
            continue
        scale = 2e-6 * max(1.0, float(o64.abs().max())) + RTOL * o64.abs()
        eh, eo = (h.double() - o64).abs() / scale, (o32.double() - o64).abs() / scale
        # (the fp32 CPU oracle itself sits 1.1x .. 4x the tolerance from float64 on this shape, depending on the host's
        # BLAS / thread count: the bound is 5x, with at most 0.5 % of the entries beyond 1x - asserted below)
        assert float(eh.max()) <= max(5.0, 1.5 * float(eo.max())), \
            f"sample {name}: HIP {float(eh.max()):.2f}x tol from float64, the fp32 CPU oracle {float(eo.max()):.2f}x"
        assert int((eh > 1.0).sum()) <= max(2, h.numel() // 200), f"sample {name}: too many ill-conditioned entries"
    # density + gradient at perturbed points (incl. points outside the tail bound and beyond the period)
    x = x_o + 0.3 * torch.randn(B, D, generator=g)
    x[0] = 7.0
    if len(circ):
        x[1, list(circ)] += 4 * math.pi
    xg = x.clone().requires_grad_(True)
    lq_o = of.log_prob(xg)
    (g_o,) = torch.autograd.grad(lq_o.sum(), xg)
    lq_h, g_h = hf.log_prob_and_grad(x.to(DEV))
    assert close(lq_h, lq_o.detach(), RTOL), f"log q: {worst(lq_h, lq_o.detach()):.2f}x tol"
    # d log q / dx is DISCONTINUOUS in x (log|dy/dx| of a C1 spline is kinked at every knot, the conditioner has ReLU
    # kinks) and the randomised 12-layer 60-D flow is stiff (|d log q / dx| up to ~3e3): fp32 itself is only good to
    # 1e-5 .. 3e-3 there (relative L2 per sample, CPU oracle against its float64 twin - and that column changes with
    # the host's BLAS threading; tools/diag_spline_grad.py prints both).  So float64 arbitrates: the HIP error
    # distribution over the samples may not be worse than the fp32 CPU oracle's own by more than 2x (median) / 3x (90th
    # percentile) beyond 1e-4; small flows additionally match the fp32 oracle element-wise on >= 90 % of the samples.
    x64 = x.double().requires_grad_(True)
    (g64,) = torch.autograd.grad(of64.log_prob(x64).sum(), x64)
    n64 = g64.norm(dim=1)
    rh = ((g_h.cpu().double() - g64).norm(dim=1) / n64).sort().values
    ro = ((g_o.double() - g64).norm(dim=1) / n64).sort().values
    for name, i, fac in (("median", B // 2, 2.0), ("90th percentile", (9 * B) // 10, 3.0)):
        assert float(rh[i]) <= fac * float(ro[i]) + 1e-4, \
            f"grad {name}: HIP {float(rh[i]):.2e} from float64, the fp32 CPU oracle {float(ro[i]):.2e}"
    assert float(rh[-1]) < 2e-2
    if D < 60:
        per_sample_ok = torch.tensor([close(g_h[b], g_o[b], RTOL, atol_scale=10) for b in range(B)])
        assert per_sample_ok.float().mean() >= 0.9, f"only {int(per_sample_ok.sum())} of {B} gradients match"
    # log_prob of the flow's own samples returns the sampling log q; autograd w.r.t. x goes through the kernels
    # (the fp32 round trip sample -> log_prob through the stiff 60-D flow inherits the inversion error bounded above)
    rt = hf.log_prob(x_h)
    assert close(rt, lq_s_h, RTOL if (D < 60 and hidden < 200) else 5e-3), f"round trip: {worst(rt, lq_s_h):.2f}x of 1e-4"
    xd = x.to(DEV).requires_grad_(True)
    (ga,) = torch.autograd.grad(hf.log_prob(xd).sum(), xd)
    assert torch.equal(ga, g_h)
    # deterministic
    lq2, g2 = hf.log_prob_and_grad(x.to(DEV))
    assert torch.equal(lq2, lq_h) and torch.equal(g2, g_h)


@pytest.mark.parametrize("D,L,hidden,circ,B", [(32, 12, 256, (), 2048), (60, 4, 256, (1, 7, 30, 59), 333), (10, 3, 250, (3,), 64),
                                               (64, 2, 256, (0, 63), 41)])
def test_the_tile_shapes_of_the_spline_density_kernels_agree(D, L, hidden, circ, B):
    """The one-launch density kernels for hidden widths padded to 256: k_spline_logprob (16 chains per workgroup on 16x16x4
    MFMAs) and k_spline_logprob_r8 (4x4x1 MFMAs, own weight image, 4, 8 or 16 chains per workgroup) differ in the summation
    order of the conditioner GEMMs only: log q to the parity tolerance, the gradient to it on all but ReLU- / knot-kink
    flips.  The three tile shapes of the 4x4x1 kernel share their arithmetic: bit-identical.  Each is deterministic and
    independent of the batch around a chain."""
    from fab_torch_amd import _ops
    of, hf = make_pair(D, L, hidden, circ, seed=D + L, std=0.2)
    g = torch.Generator().manual_seed(11)
    x = (1.5 * torch.randn(B, D, generator=g)).to(DEV)
    out = {}
    for name, mfma, shape in (("16x16x4", 16, 0), ("4x4x1 / 8 chains", 0, 8), ("4x4x1 / 16 chains", 0, 16), ("4x4x1 / 4 chains", 0, 4)):
        with _ops.option(_ops.OPT_SPLINE_MFMA, mfma), _ops.option(_ops.OPT_TILE_SHAPE, shape):
            lq, gr = hf.log_prob_and_grad(x)
            lq_only = hf.log_prob(x)
            lq_b, gr_b = hf.log_prob_and_grad(x[: B // 2 + 3].contiguous())
        assert torch.equal(lq_only, lq), f"{name}: density-only and density + gradient launches disagree"
        assert torch.equal(lq_b, lq[: B // 2 + 3]) and torch.equal(gr_b, gr[: B // 2 + 3]), f"{name}: a chain depends on its batch"
        out[name] = (lq, gr)
    a, b8, b16, b4 = out["16x16x4"], out["4x4x1 / 8 chains"], out["4x4x1 / 16 chains"], out["4x4x1 / 4 chains"]
    assert torch.equal(b8[0], b16[0]) and torch.equal(b8[1], b16[1]), "8- and 16-chain tiles of the 4x4x1 kernel differ"
    assert torch.equal(b8[0], b4[0]) and torch.equal(b8[1], b4[1]), "8- and 4-chain tiles of the 4x4x1 kernel differ"
    assert not torch.equal(a[0], b8[0]), "both runs used the same kernel"
    assert close(b8[0], a[0], RTOL), f"log q: {worst(b8[0], a[0]):.2f}x tol"
    rel = (b8[1] - a[1]).norm(dim=1) / a[1].norm(dim=1).clamp_min(1e-6)
    assert float(rel.median()) < 1e-5 and float((rel > 1e-3).float().mean()) <= 0.02, \
        f"gradient: median {float(rel.median()):.2e}, worst {float(rel.max()):.2e}"
    lq, gr = hf.log_prob_and_grad(x)                           # default: the 4x4x1 kernel, tile by batch
    assert torch.equal(lq, b8[0]) and torch.equal(gr, b8[1])


def test_a_deep_wide_spline_flow_falls_back_to_the_tile_that_fits_the_lds():
    """40 layers x 64 dimensions: the LDS plan of the 16-chain stream kernel (ReLU ballots of every layer + 4 output chunks)
    exceeds 160 KiB - the launch takes the 8-chain tile instead of failing, with the same numbers."""
    from fab_torch_amd import _ops
    D, L, hidden, B = 64, 40, 256, 24
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, (0, 5), torch.full((D,), 5.0)).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in hf.parameters():
            if p.dim() == 2 and p.shape[0] % 25 == 0:
                p.add_(0.01 * torch.randn_like(p))
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(1)).to(DEV)
    with _ops.option(_ops.OPT_TILE_SHAPE, 8):
        lq8, g8 = hf.log_prob_and_grad(x)
    with _ops.option(_ops.OPT_TILE_SHAPE, 16):
        lq16, g16 = hf.log_prob_and_grad(x)
    with _ops.option(_ops.OPT_SPLINE_MFMA, 16):
        lq0, g0 = hf.log_prob_and_grad(x)
    assert torch.isfinite(lq8).all() and torch.equal(lq8, lq16) and torch.equal(g8, g16)
    assert close(lq8, lq0, RTOL), f"log q: {worst(lq8, lq0):.2f}x tol"


def test_identity_initialised_spline_flow_is_the_base_distribution():
    D, circ = 10, (2, 5)
    tb = torch.full((D,), 5.0); tb[list(circ)] = math.pi
    hf = fa.make_wrapped_normflow_spline(D, 4, 64, circ, tb, circ_shift=None).to(DEV)
    x, lq = hf.sample_and_log_prob((256,))
    assert hf.event_shape == (D,) and x.shape == (256, D)
    base = -0.5 * math.log(2 * math.pi) * (D - 2) - 0.5 * (x[:, [i for i in range(D) if i not in circ]] ** 2).sum(1) \
        - 2 * math.log(2 * math.pi)
    assert close(lq, base, RTOL) and close(hf.log_prob(x), lq, RTOL)
    assert float(x[:, list(circ)].abs().max()) <= math.pi + 1e-5


def test_hmc_transitions_with_the_spline_flow_on_a_60d_target_vs_oracle():
    """BASELINE cfg 5's shape with the flow family it names: 60-D, spline flow 12 layers / hidden 256 / 8 bins / 12
    circular coordinates, HMC with 10 leapfrogs (OpenMM's alanine-dipeptide energy is unobtainable offline: the target
    is the 60-D ManyWell stand-in).  The flow is a non-RealNVP plug-in -> the transitions run through the generic
    path (HIP elementwise kernels + the spline kernels for the density / gradient)."""
    D, L, hidden, M, B, LF = 60, 12, 256, 4, 32, 10
    circ = (3, 7, 8, 12, 20, 21, 22, 30, 41, 45, 52, 59)
    of, hf = make_pair(D, L, hidden, circ, seed=9, std=0.2)
    target, otarget = fa.ManyWellEnergy(D), otgt.ManyWell(D)
    hop = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=LF,
                                   eval_mode=True).to(DEV)
    assert not hop.is_native
    oop = oais.HMC(M, D, of.log_prob, otarget.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=LF, eval_mode=True)
    g = torch.Generator().manual_seed(3)
    u, eps = torch.rand(B, D, generator=g), torch.randn(B, D, generator=g)
    noise_p = torch.randn(M, 1, B, D, generator=g); noise_e = torch.empty(M, 1, B).exponential_(generator=g)
    with torch.no_grad():
        x0, _ = of.sample_eps(u, eps)
    pt = oais.create_point(x0, of.log_prob, otarget.log_prob, with_grad=True)
    betas = oais.beta_schedule(M, "linear")
    for j in range(1, M + 1):
        hp = fa.Point(pt.x.clone().to(DEV), pt.log_q.clone().to(DEV), pt.log_p.clone().to(DEV),
                      pt.grad_log_q.clone().to(DEV), pt.grad_log_p.clone().to(DEV))
        lw_h = torch.zeros(B, device=DEV)
        hop.transition(hp, j, float(betas[j]), log_w=lw_h, beta_next=float(betas[j + 1]), noise_p=noise_p[j - 1].to(DEV),
                       noise_e=noise_e[j - 1].to(DEV))
        ref = oop.transition(pt.clone(), j, betas[j], noise_p[j - 1], noise_e[j - 1])
        scale = max(1.0, float(ref.x.abs().max()))
        err = (hp.x.cpu() - ref.x).abs().max(1).values / scale
        ok = err <= 1e-4
        assert int((~ok).sum()) <= 2, f"transition {j}: {int((~ok).sum())} chains differ (max {float(err.max()):.2e})"
        assert close(hp.log_q.cpu()[ok], ref.log_q[ok], RTOL) and close(hp.log_p.cpu()[ok], ref.log_p[ok], RTOL)
        lw_ref = (oais.intermediate_log_prob(ref, betas[j + 1], 2.0, False) - oais.intermediate_log_prob(ref, betas[j], 2.0, False))
        assert close(lw_h.cpu()[ok], lw_ref[ok].detach(), RTOL, atol=1e-3)
        pt = ref


def test_full_ais_call_with_the_spline_flow_as_base_distribution():
    """AnnealedImportanceSampler with the spline flow as `base_distribution` (generic plug-in path end to end): finite
    weights, nothing dropped, step sizes adapt, AIS towards p improves on plain importance sampling."""
    D, L, hidden, M, B = 6, 4, 64, 12, 2048
    circ = (1, 4)
    tb = torch.full((D,), 5.0); tb[list(circ)] = math.pi
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, circ, tb).to(DEV)          # identity-initialised: base = q0
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=True, epsilon=0.2, L=5).to(DEV)
    ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, True, None, M)
    assert not ais.is_native
    torch.manual_seed(0)
    ess_ais, ess_base = [], []
    for it in range(14):
        pt, lw = ais.sample_and_log_weights(B)
        info = ais.get_logging_info()
        if it >= 8:                                         # after the step sizes have adapted
            ess_ais.append(info["ess_ais"]); ess_base.append(info["ess_base"])
    assert pt.x.shape == (B, D) and torch.isfinite(lw).all()
    assert 0.2 < info["dist0_p_accept_0"] < 0.99
    assert sum(ess_ais) > 1.5 * sum(ess_base), f"ESS {ess_ais} (AIS) vs {ess_base} (base)"


@pytest.mark.parametrize("D,L,hidden,circ,B", [(8, 4, 64, (1, 4, 6), 100), (7, 5, 128, (0, 6), 33), (6, 3, 32, (), 64),
                                               (60, 12, 256, (3, 7, 8, 12, 20, 21, 22, 30, 41, 45, 52, 59), 64)])
def test_spline_flow_parameter_gradients_vs_oracle(D, L, hidden, circ, B):
    """Training path: d (sum_b c_b log q(x_b)) / d theta from the kernels' tape (+ one GEMM per Linear) against autograd
    through the CPU oracle, tensor by tensor under the normflows state-dict names."""
    import copy
    of, hf = make_pair(D, L, hidden, circ, seed=3 * D + L)
    hf.requires_grad_(True)
    g = torch.Generator().manual_seed(11)
    u, eps = torch.rand(B, D, generator=g), torch.randn(B, D, generator=g)
    with torch.no_grad():
        x, _ = of.sample_eps(u, eps)
    x = x + 0.2 * torch.randn(B, D, generator=g)
    c = torch.randn(B, generator=g)
    of64 = copy.deepcopy(of).double()
    for f, xx, cc in ((of, x, c), (of64, x.double(), c.double())):
        for p in f.parameters():
            p.grad = None
        (cc * f.log_prob(xx)).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    lq = hf.log_prob(xd)
    (c.to(DEV) * lq).sum().backward()
    from fab_torch_amd import _ops
    with _ops.option(_ops.OPT_SPLINE_MFMA, 16):
        lq0, gx0 = hf.log_prob_and_grad(x.to(DEV))
    # (the training forward runs the staged kernels with the tape, log_prob_and_grad the one-launch kernel ON THE SAME
    # 16x16x4 TILES: same arithmetic per coordinate, different order of the log-det row sums; the 4x4x1 stream kernel
    # that hidden 256 gets by default sums the conditioner GEMMs in another order - compared in
    # test_the_tile_shapes_of_the_spline_density_kernels_agree)
    assert close(lq.detach(), lq0, 1e-6) and close(xd.grad, c.to(DEV)[:, None] * gx0, 1e-5)
    ref32, ref64 = dict(of.named_parameters()), dict(of64.named_parameters())
    names = [n for n, _ in hf._nf_model.named_parameters()]
    assert set(names) == set(ref32) and len(names) == len(hf._train_params())
    worst_ratio = 0.0
    for n, p in hf._nf_model.named_parameters():
        assert p.grad is not None, n
        g64 = ref64[n].grad
        scale = float(g64.norm()) + 1e-30
        eh = float((p.grad.cpu().double() - g64).norm()) / scale
        eo = float((ref32[n].grad.double() - g64).norm()) / scale
        # fp32 summation over the batch and the stiff 12-layer chain: float64 arbitrates; HIP within 1e-4 relative L2 of
        # it, or no worse than 3x the fp32 CPU oracle's own distance
        assert eh <= max(1e-4, 3.0 * eo), f"{n}: HIP {eh:.2e} from float64 (fp32 CPU oracle {eo:.2e})"
        worst_ratio = max(worst_ratio, eh)
    # a second backward through a fresh forward accumulates like autograd does
    before = {n: p.grad.clone() for n, p in hf._nf_model.named_parameters()}
    (c.to(DEV) * hf.log_prob(x.to(DEV))).sum().backward()
    for n, p in hf._nf_model.named_parameters():
        assert close(p.grad, 2 * before[n], 1e-5), n


def test_spline_flow_trains_by_maximum_likelihood_on_the_gpu():
    """forward-KL training of the spline flow with torch.optim.Adam through the HIP tape: the loss must fall (the
    reference trains this family with the same optimiser, experiments/aldp/train.py)."""
    D, circ = 6, (1, 4)
    tb = torch.full((D,), 4.0); tb[list(circ)] = math.pi
    torch.manual_seed(0)
    hf = fa.make_wrapped_normflow_spline(D, 4, 64, circ, tb).to(DEV)
    g = torch.Generator().manual_seed(1)
    data = torch.randn(4096, D, generator=g) * 0.5 + 0.7
    data[:, list(circ)] = (torch.rand(4096, len(circ), generator=g) ** 2 - 0.5) * 2 * math.pi * 0.99
    data = data.to(DEV)
    opt = torch.optim.Adam(hf.parameters(), lr=3e-3)
    losses = []
    for it in range(60):
        xb = data[torch.randint(0, 4096, (512,), generator=g)]
        opt.zero_grad()
        loss = -hf.log_prob(xb).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(math.isfinite(v) for v in losses)
    assert losses[-1] < losses[0] - 1.0, f"{losses[0]:.3f} -> {losses[-1]:.3f}"


def test_fab_buffer_trainer_with_the_spline_flow():
    """The reference's alanine-dipeptide training recipe in miniature (experiments/aldp/train.py, config/fab_buff.yaml):
    spline flow + HMC-AIS towards p^2/q + prioritised buffer + torch Adam, every density / gradient / parameter gradient
    through the HIP spline kernels and the generic transition path."""
    D, L, hidden, M, B = 8, 4, 64, 4, 128
    circ = (2, 5)
    tb = torch.full((D,), 5.0); tb[list(circ)] = math.pi
    torch.manual_seed(0)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, circ, tb).to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    model = fa.FABModel(hf, target, M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    ais = model.annealed_importance_sampler
    assert not ais.is_native

    def initial_sampler():
        pt, lw = ais.sample_and_log_weights(B, logging=False)
        return pt.x, lw, pt.log_q
    buf = fa.PrioritisedReplayBuffer(D, 8 * B, 2 * B, initial_sampler, device=DEV)
    opt = torch.optim.Adam(hf.parameters(), lr=1e-3)
    trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=2.0, n_batches_buffer_sampling=2)
    before = [p.detach().clone() for p in hf.parameters()]
    x_eval = target.sample((2048,)) if hasattr(target, "sample") else None
    ll0 = float(hf.log_prob(x_eval).mean().detach()) if x_eval is not None else None
    infos = [trainer.step(i, B) for i in range(30)]
    assert all(math.isfinite(i["loss"]) and math.isfinite(i["grad_norm"]) for i in infos)
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, hf.parameters()))
    assert all(torch.isfinite(p).all() for p in hf.parameters())
    if x_eval is not None:                                  # forward KL to exact target samples improves
        ll1 = float(hf.log_prob(x_eval).mean().detach())
        assert ll1 > ll0, f"test-set log-likelihood {ll0:.3f} -> {ll1:.3f}"


def test_cfg3_with_the_spline_flow_at_full_size():
    """BASELINE cfg 3 as BASELINE.json words it: ManyWell-32, SPLINE flow 12 layers (hidden 256, 8 bins), 2048 chains, 12
    intermediate distributions, HMC(5), prioritised replay buffer - one trainer iteration at full size through the
    generic plug-in path; the flow's densities of a 64-chain slice of the AIS batch against the CPU oracle."""
    D, L, hidden, M, B, SL = 32, 12, 256, 12, 2048, 64
    torch.manual_seed(0)
    of = osp.make_circular_coupled_flow(D, L, hidden, (), torch.full((D,), 5.0), seed=0)
    osp.randomize(of, 0.1, 1)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, (), 5.0, seed=0)
    hf._nf_model.load_state_dict(of.state_dict(), strict=True)
    hf = hf.to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=5).to(DEV)
    model = fa.FABModel(hf, target, M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    ais = model.annealed_importance_sampler

    def initial_sampler():
        pt, lw = ais.sample_and_log_weights(B, logging=False)
        return pt.x, lw, pt.log_q
    buf = fa.PrioritisedReplayBuffer(D, 6 * B, 2 * B, initial_sampler, device=DEV)
    opt = torch.optim.Adam(hf.parameters(), lr=1e-4)
    trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=2.0, n_batches_buffer_sampling=2)
    # the AIS batch of a fresh call: point fields consistent with the oracle flow / target on a slice
    pt, lw = ais.sample_and_log_weights(B)
    assert pt.x.shape == (B, D) and torch.isfinite(lw).all() and torch.isfinite(pt.x).all()
    xs = pt.x[:SL].cpu()
    with torch.no_grad():
        lq_o = of.log_prob(xs)
    assert close(pt.log_q[:SL], lq_o, RTOL), f"log q of the AIS points: {worst(pt.log_q[:SL], lq_o):.2f}x tol"
    assert close(pt.log_p[:SL], otgt.ManyWell(D).log_prob(xs), RTOL)
    before = [p.detach().clone() for p in hf.parameters()]
    info = trainer.step(0, B)
    assert math.isfinite(info["loss"]) and math.isfinite(info["grad_norm"]) and 0 < info["ess_ais"] <= 1
    idx = trainer.last_indices
    assert idx.shape == (2 * B,) and len(set(idx.tolist())) == 2 * B
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, hf.parameters()))
    assert all(torch.isfinite(p).all() for p in hf.parameters())


def test_eval_info_with_the_spline_flow():
    """FABModel.get_eval_info / AnnealedImportanceSampler.generate_eval_data (ais.py:132-188, core.py:191-220) through the
    generic path: base samples carry log p - log q of the sampling pass, AIS towards p, ManyWell metrics on both."""
    D, L, hidden, M = 6, 3, 64, 3
    torch.manual_seed(0)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, (), 5.0).to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    model = fa.FABModel(hf, target, M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    bx, blw, ax, alw = model.annealed_importance_sampler.generate_eval_data(512, 256)
    assert bx.shape == (512, D) and ax.shape == (512, D) and blw.shape == (512,) and alw.shape == (512,)
    with torch.no_grad():
        assert close(blw, target.log_prob(bx) - hf.log_prob(bx), 1e-4)            # log p - log q at the base samples
    info = model.get_eval_info(outer_batch_size=512, inner_batch_size=256)
    for k in ("eval_ess_flow", "eval_ess_ais", "flow_forward_kl", "flow_test_set_exact_mean_log_prob",
              "ais_abs_MSE_log_Z_estimate"):
        assert k in info and math.isfinite(info[k]), k
    assert 0 < info["eval_ess_flow"] <= 1 and 0 < info["eval_ess_ais"] <= 1
    assert model.annealed_importance_sampler.p_target is False                      # toggled back (core.py:219)


def test_plain_trainer_forward_kl_and_checkpoint_roundtrip_with_the_spline_flow(tmp_path):
    """fab/train.py's Trainer (model.loss -> backward -> clip -> step) with the spline flow for both losses it can take
    through the HIP tape (`fab_alpha_div`, `target_forward_kl`), and FABModel.save / load (core.py:222-251) incl. a raw
    `_nf_model` state dict as a normflows checkpoint would be."""
    D, L, hidden, M, B = 6, 3, 64, 3, 128
    torch.manual_seed(0)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, (1,), torch.tensor([5.0, math.pi, 5.0, 5.0, 5.0, 5.0])).to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    for loss_type in ("fab_alpha_div", "target_forward_kl"):
        model = fa.FABModel(hf, target, M, alpha=2.0, transition_operator=hmc, loss_type=loss_type)
        opt = torch.optim.Adam(hf.parameters(), lr=1e-3)
        trainer = fa.Trainer(model, opt, save_path=str(tmp_path))
        before = [p.detach().clone() for p in hf.parameters()]
        infos = [trainer.step(i, B) for i in range(3)]
        assert all(math.isfinite(i["loss"]) and math.isfinite(i["grad_norm"]) for i in infos), loss_type
        assert any(not torch.equal(a, p.detach()) for a, p in zip(before, hf.parameters())), loss_type
    path = str(tmp_path / "model.pt")
    model.save(path)
    x = target.sample((64,))
    ref = hf.log_prob(x).detach()
    hf2 = fa.make_wrapped_normflow_spline(D, L, hidden, (1,), torch.tensor([5.0, math.pi, 5.0, 5.0, 5.0, 5.0])).to(DEV)
    hmc2 = fa.HamiltonianMonteCarlo(M, D, hf2.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    model2 = fa.FABModel(hf2, target, M, alpha=2.0, transition_operator=hmc2, loss_type="fab_alpha_div")
    assert not torch.equal(hf2.log_prob(x).detach(), ref)
    model2.load(path, map_location=DEV)
    assert torch.equal(hf2.log_prob(x).detach(), ref)
    assert torch.equal(hmc2.epsilons, hmc.epsilons) and torch.equal(hmc2.common_epsilon, hmc.common_epsilon)
    raw = {"flow": hf._nf_model.state_dict(), "trans_op": hmc.state_dict()}     # keys as normflows itself would save them
    torch.save(raw, str(tmp_path / "raw.pt"))
    hf3 = fa.make_wrapped_normflow_spline(D, L, hidden, (1,), torch.tensor([5.0, math.pi, 5.0, 5.0, 5.0, 5.0])).to(DEV)
    hmc3 = fa.HamiltonianMonteCarlo(M, D, hf3.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    fa.FABModel(hf3, target, M, alpha=2.0, transition_operator=hmc3, loss_type="fab_alpha_div").load(str(tmp_path / "raw.pt"), DEV)
    assert torch.equal(hf3.log_prob(x).detach(), ref)


def test_flat_adam_on_the_spline_flow_matches_clip_grad_norm_and_torch_adam():
    """FlatAdam generalised to any module (parameters re-pointed into one flat buffer, gradients concatenated, ONE fused
    clip + Adam launch): the buffer trainer with it follows the torch.optim.Adam + clip_grad_norm_ run step for step."""
    D, L, hidden, M, B = 6, 3, 64, 3, 128
    tb = torch.tensor([5.0, math.pi, 5.0, 5.0, 5.0, 5.0])
    runs = {}
    for kind in ("torch", "flat"):
        torch.manual_seed(0)
        hf = fa.make_wrapped_normflow_spline(D, L, hidden, (1,), tb).to(DEV)
        target = fa.ManyWellEnergy(D)
        hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
        model = fa.FABModel(hf, target, M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
        ais = model.annealed_importance_sampler

        def initial_sampler():
            pt, lw = ais.sample_and_log_weights(B, logging=False)
            return pt.x, lw, pt.log_q
        torch.manual_seed(1)
        buf = fa.PrioritisedReplayBuffer(D, 8 * B, 2 * B, initial_sampler, device=DEV)
        opt = fa.FlatAdam(hf, lr=1e-3) if kind == "flat" else torch.optim.Adam(hf.parameters(), lr=1e-3)
        if kind == "flat":
            assert not opt.native and opt.theta.numel() == sum(p.numel() for p in hf.parameters())
        trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=2.0, n_batches_buffer_sampling=2, max_gradient_norm=1.0)
        torch.manual_seed(2)
        infos = [trainer.step(i, B) for i in range(3)]
        runs[kind] = ({k: v.detach().clone() for k, v in hf.state_dict().items()}, infos)
    for k, v in runs["torch"][0].items():
        assert close(runs["flat"][0][k], v, 1e-4, atol_scale=50), k
    for a, b in zip(runs["torch"][1], runs["flat"][1]):
        assert abs(a["loss"] - b["loss"]) <= 1e-4 * max(1.0, abs(a["loss"])) and abs(a["grad_norm"] - b["grad_norm"]) <= 1e-3 * a["grad_norm"]


@pytest.mark.parametrize("n_outer", [1, 2])
def test_host_fused_spline_transition_equals_the_step_by_step_generic_path(monkeypatch, n_outer):
    """`fabhip::spline_hmc_transition` enqueues exactly the launches the Python loop of `_transition_generic` issues (one op
    call per transition instead of ~40): same state, weights, step sizes and logging slots, bit for bit."""
    D, L, hidden, M, B, LF = 8, 3, 64, 3, 200, 4
    tb = torch.full((D,), 5.0); tb[[1, 6]] = math.pi
    torch.manual_seed(0)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, (1, 6), tb).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in hf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    res = {}
    for mode in ("fused", "stepwise"):
        if mode == "stepwise":
            monkeypatch.setattr(fa.HamiltonianMonteCarlo, "force_stepwise", True)
        hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.15, L=LF,
                                       n_outer=n_outer).to(DEV)
        g = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(B, D, device=DEV, generator=g)
        pt = hmc.create_new_point(x)
        lw = torch.zeros(B, device=DEV)
        for j in (1, 2, 3):
            npz = torch.randn(n_outer, B, D, device=DEV, generator=g)
            nez = torch.empty(n_outer, B, device=DEV).exponential_(generator=g)
            hmc.transition(pt, j, 0.25 * j, log_w=lw, beta_next=0.25 * (j + 1), noise_p=npz, noise_e=nez)
        res[mode] = (pt.x.clone(), pt.log_q.clone(), pt.grad_log_p.clone(), lw.clone(), hmc.epsilons.clone(),
                     hmc.common_epsilon.clone(), hmc.get_logging_info())
    for a, b in zip(res["fused"][:6], res["stepwise"][:6]):
        assert torch.equal(a, b)
    assert res["fused"][6] == res["stepwise"][6]
    assert not torch.equal(res["fused"][3], torch.zeros(B, device=DEV))


@pytest.mark.parametrize("D,L,hidden,circ,B,n_outer", [(8, 3, 64, (1, 6), 200, 1), (32, 4, 256, (), 130, 2)])
def test_fused_spline_ais_call_equals_the_step_by_step_generic_path(monkeypatch, D, L, hidden, circ, B, n_outer):
    """`fabhip_spline_ais_run` (flow sample, point creation, initial weights, "chain init" filter, base ESS,
    M HMC transitions, "chain end" filter, ESS / log Z in ONE op call, like fabhip_ais_run for the RealNVP family) against the
    reference's loop stepped from Python over the same kernels (`_sample_generic`): particles, log-weights, adapted step sizes
    and the logging scalars bit for bit; a chain whose flow sample is non-finite is dropped by both."""
    M, LF = 4, 3
    tb = torch.full((D,), 5.0)
    if circ:
        tb[list(circ)] = math.pi
    torch.manual_seed(0)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, circ, tb).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in hf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(9)
    na = torch.randn(M, n_outer, B, D, device=DEV, generator=g)
    nb = torch.empty(M, n_outer, B, device=DEV).exponential_(generator=g)
    res = {}
    for mode in ("fused", "stepwise"):
        monkeypatch.setattr(fa.HamiltonianMonteCarlo, "force_stepwise", mode == "stepwise")
        hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.15, L=LF,
                                       n_outer=n_outer).to(DEV)
        ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
        assert (ais._spline_parts() is not None) == (mode == "fused")
        torch.manual_seed(21)                               # the flow's base draws: u = rand, eps = randn, in that order
        pt, lw = ais.sample_and_log_weights(B, noise_a=na, noise_b=nb)
        res[mode] = (pt.x.clone(), pt.log_q.clone(), pt.log_p.clone(), pt.grad_log_q.clone(), lw.clone(),
                     hmc.epsilons.clone(), hmc.common_epsilon.clone(), ais.get_logging_info())
    for a, b in zip(res["fused"][:7], res["stepwise"][:7]):
        assert a.shape == b.shape and torch.equal(a, b)
    fi, si = res["fused"][7], res["stepwise"][7]
    assert fi.keys() == si.keys()
    for k in fi:
        assert abs(fi[k] - si[k]) <= 1e-6 * max(1.0, abs(si[k])), (k, fi[k], si[k])
    assert res["fused"][0].shape[0] == B and not torch.equal(res["fused"][5], torch.full_like(res["fused"][5], 0.15 * 0.9))


@pytest.mark.parametrize("name,D,L,hidden,circ,B,M,LF,shape", [
    ("cfg3", 32, 12, 256, (), 2048, 3, 5, 8),
    ("cfg5-shape", 60, 12, 256, (3, 7, 8, 12, 20, 21, 22, 30, 41, 45, 52, 59), 4096, 2, 10, 16)])
def test_spline_transitions_at_the_baseline_tile_shapes_vs_oracle(name, D, L, hidden, circ, B, M, LF, shape):
    """the spline family's transitions AT THE TILE SHAPES THE BASELINE BATCHES SELECT - cfg 3: 2048 chains =
    8-chain tiles, cfg 5's shape: 4096 chains = 16-chain stream tiles, one launch per leapfrog (half steps + target inside the
    density kernel) - against the CPU oracle per transition: a FULL-SIZE batch whose first 32 rows are the oracle's state
    (teacher-forced), the other rows filler chains; at most 2 of the 32 may differ (an accept decision / ReLU kink within
    rounding), everything else within 1e-4.  Then the fused AIS call at full size: its first 32 chains are, bit for bit, what
    a 32-chain call at the same tile shape gives on the same noise rows (chains do not depend on the batch they run in)."""
    from fab_torch_amd import _ops
    SL = 32
    of, hf = make_pair(D, L, hidden, circ, seed=11, std=0.15)
    target, otarget = fa.ManyWellEnergy(D), otgt.ManyWell(D)
    hop = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=LF,
                                   eval_mode=True).to(DEV)
    oop = oais.HMC(M, D, of.log_prob, otarget.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=LF, eval_mode=True)
    g = torch.Generator().manual_seed(5)
    u, eps = torch.rand(B, D, generator=g), torch.randn(B, D, generator=g)
    noise_p = torch.randn(M, 1, B, D, generator=g); noise_e = torch.empty(M, 1, B).exponential_(generator=g)
    with torch.no_grad():
        x0, _ = of.sample_eps(u[:SL], eps[:SL])
    filler, _ = hf.sample_and_log_prob((B - SL,))
    pt = oais.create_point(x0, of.log_prob, otarget.log_prob, with_grad=True)
    betas = oais.beta_schedule(M, "linear")
    assert shape == (8 if B <= 8 * 256 else 16)            # what r8_row_blocks picks for this batch on a 256-CU device
    for j in range(1, M + 1):
        xfull = torch.cat([pt.x.to(DEV), filler], 0)
        hp = hop.create_new_point(xfull)
        assert close(hp.log_q[:SL], pt.log_q, RTOL) and max_rel_err(hp.grad_log_q[:SL], pt.grad_log_q) <= 5e-4
        lw_h = torch.zeros(B, device=DEV)
        hop.transition(hp, j, float(betas[j]), log_w=lw_h, beta_next=float(betas[j + 1]), noise_p=noise_p[j - 1].to(DEV),
                       noise_e=noise_e[j - 1].to(DEV))
        ref = oop.transition(pt.clone(), j, betas[j], noise_p[j - 1][:, :SL], noise_e[j - 1][:, :SL])
        scale = max(1.0, float(ref.x.abs().max()))
        err = (hp.x[:SL].cpu() - ref.x).abs().max(1).values / scale
        ok = err <= 1e-4
        assert int((~ok).sum()) <= 2, f"{name} transition {j}: {int((~ok).sum())} chains differ (max {float(err.max()):.2e})"
        assert close(hp.log_q[:SL].cpu()[ok], ref.log_q[ok], RTOL) and close(hp.log_p[:SL].cpu()[ok], ref.log_p[ok], RTOL)
        lw_ref = (oais.intermediate_log_prob(ref, betas[j + 1], 2.0, False) - oais.intermediate_log_prob(ref, betas[j], 2.0, False))
        assert close(lw_h[:SL].cpu()[ok], lw_ref[ok].detach(), RTOL, atol=1e-3)
        assert torch.isfinite(hp.x).all()
        filler = hp.x[SL:].clone()
        pt = ref
    # the fused call: full size vs its first 32 chains alone at the same tile shape
    ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hop, False, 2.0, M)
    assert ais._spline_parts() is not None
    na, nb = noise_p.to(DEV), noise_e.to(DEV)
    full, lw_full = ais.sample_and_log_weights(B, eps0=eps.to(DEV), u0=u.to(DEV), noise_a=na, noise_b=nb)
    assert full.x.shape[0] == B, "a chain was dropped: pick another seed"
    with _ops.option(_ops.OPT_TILE_SHAPE, shape):
        part, lw_part = ais.sample_and_log_weights(SL, eps0=eps[:SL].to(DEV), u0=u[:SL].to(DEV),
                                                   noise_a=na[:, :, :SL].contiguous(), noise_b=nb[:, :, :SL].contiguous())
    assert torch.equal(full.x[:SL], part.x) and torch.equal(lw_full[:SL], lw_part) and torch.equal(full.log_q[:SL], part.log_q)


@pytest.mark.parametrize("D,L,hidden,circ,B,n_outer,LF,shape", [
    (32, 4, 256, (), 2041, 1, 3, 8),                       # cfg 3's tile shape, a ragged last workgroup AND a ragged last 16-row block
    (32, 3, 256, (), 301, 2, 1, 4),                        # 4-chain tiles; L = 1: first and last leapfrog are the same launch; two outer steps
    (60, 3, 256, (3, 7, 20), 530, 1, 2, 16),               # cfg 5's shape: 16-chain stream tiles, D > 32 (four coordinates per lane of a row)
])
def test_spline_transition_in_L_launches_equals_the_L_plus_3_launch_form(D, L, hidden, circ, B, n_outer, LF, shape):
    """Round 5 (VERDICT r4 Missing #2): with FABHIP_ADAPT_FOLD the fused spline call runs an outer HMC step in L launches - the
    first leapfrog launch does k_gen_hmc_begin's work at its top, the last one k_gen_hmc_accept's and (in the last wave of
    the launch to finish) k_gen_hmc_adapt's.  Same sums in the same order: particles, densities, gradients, log-weights,
    ADAPTED step sizes (tuning on) and the logging scalars must equal the L + 3 launch form bit for bit - on every tile
    shape, with chains dropped by the "chain init" filter (rows past the device-side row count) and ragged batch sizes."""
    from fab_torch_amd import _ops
    M = 3
    tb = torch.full((D,), 5.0)
    if circ:
        tb[list(circ)] = math.pi
    torch.manual_seed(3)
    hf = fa.make_wrapped_normflow_spline(D, L, hidden, circ, tb).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in hf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(4)
    na = torch.randn(M, n_outer, B, D, device=DEV, generator=g)
    nb = torch.empty(M, n_outer, B, device=DEV).exponential_(generator=g)
    u0 = torch.rand(B, D, device=DEV, generator=g)
    eps0 = torch.randn(B, D, device=DEV, generator=g)
    eps0[5, 0] = float("nan")                               # one chain leaves at the "chain init" filter
    res = {}
    for fold in (1, 0):
        with _ops.option(_ops.OPT_TILE_SHAPE, shape), _ops.option(_ops.OPT_ADAPT_FOLD, fold):
            hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.12, L=LF,
                                           n_outer=n_outer).to(DEV)
            ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
            assert ais._spline_parts() is not None
            pt, lw = ais.sample_and_log_weights(B, noise_a=na, noise_b=nb, u0=u0, eps0=eps0)
            res[fold] = (pt.x.clone(), pt.log_q.clone(), pt.log_p.clone(), pt.grad_log_q.clone(), pt.grad_log_p.clone(), lw.clone(),
                         hmc.epsilons.clone(), hmc.common_epsilon.clone(), ais.get_logging_info())
    for a, b in zip(res[1][:8], res[0][:8]):
        assert a.shape == b.shape and torch.equal(a, b)
    assert res[1][8] == res[0][8]
    assert res[1][0].shape[0] == B - 1
    assert not torch.equal(res[1][6], torch.full_like(res[1][6], 0.12 * 0.9))        # the rule ran
