"""Per-transition HMC / Metropolis parity (HIP vs CPU oracle, identical inputs and noise) on the kernel shapes the
golden fixtures do not reach: 8 column tiles per wave (W = 512), D > 32 (4 k-blocks for the D x D maps and the first
reverse GEMM: the streaming code paths), narrow flows.  Same acceptance rule as the headline test: 1e-4 of the state
scale, at most one chain per transition may differ through an accept decision within rounding of the threshold."""
import pytest
import torch

from helpers import close, max_rel_err, RTOL
from test_gpu_parity import seeded_flow, hip_flow_from_oracle, DEV

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from fab_torch_amd import _ops            # noqa: E402
from oracle import ais as oais            # noqa: E402
from oracle import targets as otgt        # noqa: E402


@pytest.mark.parametrize("D,K,nodes,M,L,B", [(64, 2, 8, 3, 3, 24), (60, 3, 4, 2, 4, 20), (6, 8, 40, 3, 5, 40),
                                             (32, 2, 16, 2, 3, 16), (4, 2, 4, 3, 2, 33)])
def test_hmc_transitions_vs_oracle_on_other_kernel_shapes(D, K, nodes, M, L, B):
    nf = seeded_flow(D, K, nodes, 40 + D + K)
    hf = hip_flow_from_oracle(nf)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.12, L=L,
                                   eval_mode=True).to(DEV)
    ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
    torch.manual_seed(13)
    eps0 = torch.randn(B, D)
    noise_p = torch.randn(M, 1, B, D)
    noise_e = torch.empty(M, 1, B).exponential_()
    otarget = otgt.ManyWell(D)
    ohmc = oais.HMC(M, D, nf.log_prob, otarget.log_prob, alpha=2.0, p_target=False, epsilon=0.12, L=L, eval_mode=True)
    oa = oais.AIS(lambda e: tuple(t.detach() for t in nf.sample_eps(e)), nf.log_prob, otarget.log_prob, ohmc, False,
                  2.0, M)
    oa.sample_and_log_weights(eps0, noise_p, noise_e, keep_snapshots=True)
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
        flipped = err > 1e-4
        assert flipped.sum() <= 1, f"transition {j}: {int(flipped.sum())} chains differ (max err {float(err.max()):.2e})"
        ok = ~flipped
        assert close(lw.cpu()[ok], lw_ref[ok], RTOL), f"transition {j}: log_w err {max_rel_err(lw.cpu()[ok], lw_ref[ok]):.2e}"
        assert close(pt.log_q.cpu()[ok], p_ref.log_q[ok], RTOL)
        assert close(pt.grad_log_q.cpu()[ok], p_ref.grad_log_q[ok], 5e-4)
        assert close(pt.log_p.cpu()[ok], p_ref.log_p[ok], RTOL)


@pytest.mark.parametrize("D,K,nodes", [(32, 10, 10), (60, 2, 4)])
def test_hmc_transition_is_bitwise_deterministic_over_repeated_launches(D, K, nodes):
    """Race screen for the fused transition kernel (hand-counted vmcnt ring + compiler-tracked early loads + LDS
    hand-offs without a workgroup barrier): 257 workgroups, identical inputs and noise, 12 launches must agree bit
    for bit, and a subsample must match the oracle."""
    B, M, L = 257 * 16, 2, 3
    nf = seeded_flow(D, K, nodes, 70 + D)
    hf = hip_flow_from_oracle(nf)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=L,
                                   eval_mode=True).to(DEV)
    torch.manual_seed(21)
    x0 = hf.native_sample(torch.randn(B, D, device=DEV))[0]
    noise_p = torch.randn(1, B, D, device=DEV)
    noise_e = torch.empty(1, B, device=DEV).exponential_()
    from fab_torch_amd.transition_operators import create_point
    outs = []
    for _ in range(12):
        pt = create_point(x0.clone(), hf, target, with_grad=True)
        lw = torch.zeros(B, device=DEV)
        hmc.transition(pt, 1, 1.0 / 3, log_w=lw, beta_next=2.0 / 3, noise_p=noise_p, noise_e=noise_e)
        outs.append((pt.x.clone(), pt.log_q.clone(), pt.grad_log_q.clone(), lw.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    # oracle on a subsample of chains spread over the workgroups
    idx = torch.arange(0, B, 173)
    otarget = otgt.ManyWell(D)
    ohmc = oais.HMC(M, D, nf.log_prob, otarget.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=L, eval_mode=True)
    op = oais.create_point(x0[idx].cpu(), nf.log_prob, otarget.log_prob, with_grad=True)
    op = ohmc.transition(op, 1, 1.0 / 3, noise_p[:, idx].cpu(), noise_e[:, idx].cpu())
    err = (outs[0][0][idx].cpu() - op.x).abs().max(1).values / max(1.0, float(op.x.abs().max()))
    assert (err > 1e-4).sum() <= 1, f"{int((err > 1e-4).sum())} of {len(idx)} sampled chains differ from the oracle"


@pytest.mark.parametrize("D,K,nodes,B,small", [(32, 10, 10, 1024, 4), (6, 3, 8, 70, 4), (60, 4, 4, 200, 4), (2, 2, 40, 33, 4),
                                               (16, 3, 16, 257, 4), (32, 10, 10, 2048, 8), (16, 3, 16, 257, 8),
                                               (32, 12, 8, 1501, 8), (8, 2, 32, 29, 8), (32, 3, 9, 100, 8), (20, 2, 13, 50, 8), (6, 2, 40, 33, 8)])
def test_small_tiles_match_sixteen_chain_tiles(monkeypatch, D, K, nodes, B, small):
    """k_hmc_step_r4 (4 chains per workgroup on v_mfma_f32_4x4x1, used for B <= 1152) and k_hmc_step_r8 (8 chains per
    workgroup, one wave per 64 hidden columns: D <= 32, hidden width 193 .. 320, used for 1152 < B <= 2048) against
    k_hmc_step (16 chains per workgroup): the same transition from the same point and noise.  The two differ only in the summation order inside
    the GEMMs, so per-chain results agree to fp32 rounding except where an accept / reject decision flips; the
    acceptance sums feeding the step-size rule are added in the same order by construction (bit-equal epsilons
    whenever no decision flipped)."""
    torch.manual_seed(D * 100 + K)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    target = fa.GMM(D, n_mixes=5, loc_scaling=2.0, seed=1, true_expectation_estimation_n_samples=1000) if D == 2 else \
        (fa.ManyWellEnergy(D) if D % 2 == 0 else None)
    if target is None:
        pytest.skip("ManyWell needs an even dimension")
    res = {}
    for mode in ("0", "1"):
        _ops.load().set_option(_ops.OPT_TILE_SHAPE, small if mode == "1" else 16)
        hmc = fa.HamiltonianMonteCarlo(4, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                       n_outer=2, L=3).to(DEV)
        g = torch.Generator(device=DEV).manual_seed(5)
        x0, _ = flow.native_sample(torch.randn(B, D, device=DEV, generator=g))
        pt = fa.create_point(x0, flow, target, with_grad=True)
        torch.manual_seed(77)
        out = hmc.transition(pt, 2, 0.4)
        res[mode] = (out.x.clone(), out.log_q.clone(), out.log_p.clone(), hmc.epsilons.clone(), hmc.common_epsilon.clone())
    _ops.load().set_option(_ops.OPT_TILE_SHAPE, 0)
    xa, lqa, lpa, ea, ca = res["0"]
    xb, lqb, lpb, eb, cb = res["1"]
    same = (xa - xb).abs().amax(dim=1) <= 1e-4 * (1 + xa.abs().amax(dim=1))
    assert float(same.float().mean()) >= 0.98                         # a flipped decision moves a whole chain
    assert float((lqa[same] - lqb[same]).abs().max()) <= 2e-3 * (1 + float(lqa[same].abs().max()))
    assert torch.allclose(ea, eb, rtol=0, atol=0) or float(same.float().mean()) < 1.0
    assert torch.allclose(ca, cb, rtol=1e-6)
    if small == 8 or D <= 32:
        assert not torch.equal(lqa, lqb), "both runs used the same kernel"


def test_eight_chain_tiles_are_bitwise_reproducible_and_independent_of_the_batch():
    """Race screen for k_hmc_step_r8 / k_ais_init_r8 (weight ring in AGPRs filled by inline-asm loads, LDS-only barriers):
    the same AIS call on the same noise five times at 2048 chains (256 workgroups of 5 waves) and at a ragged 1499:
    identical bits; and a chain's result does not depend on the chains around it."""
    D, K, nodes, M = 32, 10, 10, 4
    torch.manual_seed(1)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    with _ops.option(_ops.OPT_TILE_SHAPE, 8):
        for B in (2048, 1499):
            outs = []
            for rep in range(5):
                hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1,
                                               n_outer=1, L=5, eval_mode=True).to(DEV)
                ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
                g = torch.Generator().manual_seed(3)
                eps0 = torch.randn(B, D, generator=g).to(DEV)
                na = torch.randn(M, 1, B, D, generator=g).to(DEV)
                nb = torch.empty(M, 1, B).exponential_(generator=g).to(DEV)
                pt, lw = ais.sample_and_log_weights(B, eps0=eps0, noise_a=na, noise_b=nb)
                outs.append((pt.x.clone(), lw.clone()))
            for o in outs[1:]:
                assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
            n = 517                                          # the first 517 chains alone (step sizes frozen: eval mode)
            pt, lw = ais.sample_and_log_weights(n, eps0=eps0[:n].contiguous(), noise_a=na[:, :, :n].contiguous(),
                                                noise_b=nb[:, :, :n].contiguous())
            assert torch.equal(pt.x, outs[0][0][:n]) and torch.equal(lw, outs[0][1][:n])


def test_four_chain_tiles_are_bitwise_reproducible():
    """Race screen for k_hmc_step_r4 (LDS-only barriers written in inline asm, weight requests in flight across them):
    the same AIS call on the same noise, five times at 1024 chains (256 workgroups) and at a ragged 1027: identical bits."""
    D, K, nodes, M = 32, 10, 10, 4
    torch.manual_seed(1)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    for B in (1024, 1027):
        outs = []
        for rep in range(5):
            hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1,
                                           n_outer=1, L=5).to(DEV)
            ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
            torch.manual_seed(123)
            pt, lw = ais.sample_and_log_weights(B)
            outs.append((pt.x.clone(), lw.clone(), hmc.epsilons.clone()))
        for o in outs[1:]:
            assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


@pytest.mark.parametrize("D,K,nodes,B", [(32, 10, 8, 1024), (6, 3, 40, 70), (16, 3, 8, 257)])
def test_stream_and_staged_four_chain_kernels_are_bit_identical(monkeypatch, D, K, nodes, B):
    """The 4-chain kernel exists in two request schedules - one continuous weight stream per wave (D <= 32, hidden width
    >= 128) and per-stage request groups (everything else; FABHIP_OPT_R4_STREAM = 0 forces it) - with the same arithmetic in the
    same order: identical bits."""
    torch.manual_seed(D + K)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    res = []
    g = torch.Generator(device=DEV).manual_seed(5)
    x0, _ = flow.native_sample(torch.randn(B, D, device=DEV, generator=g))     # (one starting point: the flow SAMPLE has a 4-chain
    for mode in ("1", "0"):                                                    #  form on the stream image only)
        _ops.load().set_option(_ops.OPT_R4_STREAM, int(mode))
        hmc = fa.HamiltonianMonteCarlo(3, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                       n_outer=2, L=4).to(DEV)
        pt = fa.create_point(x0.clone(), flow, target, with_grad=True)        # (the transition commits in place)
        torch.manual_seed(7)
        out = hmc.transition(pt, 1, 0.3)
        res.append((out.x.clone(), out.log_q.clone(), out.grad_log_q.clone(), hmc.epsilons.clone()))
    _ops.load().set_option(_ops.OPT_R4_STREAM, 2)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D,K,nodes,B", [(32, 10, 10, 1024), (32, 10, 8, 37), (6, 3, 40, 70), (16, 3, 8, 257), (10, 2, 16, 9),
                                         (32, 4, 4, 64)])
def test_fused_stage_four_chain_kernels_match_the_stream_kernels_and_sixteen_chain_tiles(D, K, nodes, B):
    """flow_r4f.h (FABHIP_OPT_R4_STREAM = 2, the default): the D x D map multiplied in one stage with the first / last conditioner
    Linear (W1' = W'[:, :d] W1^T formed in float64 at pack time), the coupling in the W3 epilogue.  Same function, other rounding:
    one HMC transition (two outer steps) and the flow sample agree with the round-3 stream kernels (option 1) and with the
    16-chain tiles to 1e-5 of the scale wherever no accept decision flipped, and the fused kernels are bit-reproducible."""
    torch.manual_seed(D + K)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.03 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(5)
    eps = torch.randn(B, D, device=DEV, generator=g)
    ops = _ops.load()
    res, smp = {}, {}
    try:
        for mode, shape in ((2, 4), (2, 4), (1, 4), (2, 16)):
            ops.set_option(_ops.OPT_R4_STREAM, mode)
            ops.set_option(_ops.OPT_TILE_SHAPE, shape)
            xs, lqs = flow.native_sample(eps)
            smp.setdefault((mode, shape), []).append((xs.clone(), lqs.clone()))
        x0 = smp[(2, 16)][0][0]                                        # one starting point for every variant
        for mode, shape in ((2, 4), (2, 4), (1, 4), (2, 16)):
            ops.set_option(_ops.OPT_R4_STREAM, mode)
            ops.set_option(_ops.OPT_TILE_SHAPE, shape)
            hmc = fa.HamiltonianMonteCarlo(3, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                           n_outer=2, L=4).to(DEV)
            pt = fa.create_point(x0.clone(), flow, target, with_grad=True)
            torch.manual_seed(7)
            out = hmc.transition(pt, 1, 0.3)
            res.setdefault((mode, shape), []).append((out.x.clone(), out.log_q.clone(), out.grad_log_q.clone(), hmc.epsilons.clone()))
    finally:
        ops.set_option(_ops.OPT_R4_STREAM, 2)
        ops.set_option(_ops.OPT_TILE_SHAPE, 0)
    # bit-reproducible
    for a, b in zip(res[(2, 4)][0], res[(2, 4)][1]):
        assert torch.equal(a, b)
    for a, b in zip(smp[(2, 4)][0], smp[(2, 4)][1]):
        assert torch.equal(a, b)
    # the sample: x and log q against the round-3 stream and the 16-chain sampler
    for other in ((1, 4), (2, 16)):
        xa, la = smp[(2, 4)][0]
        xb, lb = smp[other][0]
        sc = max(1.0, float(xb.abs().max()))
        assert float((xa - xb).abs().max()) <= 1e-5 * sc, (other, float((xa - xb).abs().max()))
        assert float((la - lb).abs().max()) <= 1e-5 * max(1.0, float(lb.abs().max()))
    # the transition
    for other in ((1, 4), (2, 16)):
        xa, lqa, ga, ea = res[(2, 4)][0]
        xb, lqb, gb, eb = res[other][0]
        sc = max(1.0, float(xb.abs().max()))
        err = (xa - xb).abs().max(1).values / sc
        ok = err <= 1e-5
        assert int((~ok).sum()) <= max(1, B // 200), (other, int((~ok).sum()), float(err.max()))
        assert float((lqa[ok] - lqb[ok]).abs().max()) <= 2e-5 * max(1.0, float(lqb.abs().max()))
        # (the gradient of a ReLU network jumps where a pre-activation changes sign: a chain that ends within rounding of a kink
        #  may show an O(1e-3) gradient difference at an O(1e-6) difference in x - counted, not excluded silently)
        gerr = (ga - gb).abs().max(1).values / max(1.0, float(gb.abs().max()))
        assert int(((gerr > 1e-4) & ok).sum()) <= max(1, B // 100), (other, int(((gerr > 1e-4) & ok).sum()), float(gerr.max()))
        if bool(ok.all()):
            assert torch.equal(ea, eb)


@pytest.mark.parametrize("D,K,nodes,B", [(32, 10, 10, 2048), (32, 10, 8, 37), (6, 3, 40, 70), (16, 3, 20, 257), (10, 2, 30, 9)])
def test_fused_stage_eight_chain_kernels_match_the_unfused_ones_and_sixteen_chain_tiles(D, K, nodes, B):
    """flow_r8.h with fused stages (FABHIP_OPT_R4_STREAM >= 2, the default; hidden widths 256 / 320): y -> z and y -> h1 in ONE
    stage (W1' = W'[:, :d] W1^T in float64 at pack time), dh1 W1'^T + g_z W'^T as ONE K-split stage.  One HMC transition (two outer
    steps) agrees with the unfused 8-chain kernel (option 1) and the 16-chain tiles to 1e-5 of the scale wherever no accept decision
    flipped; bit-reproducible; the initial density of a chain (k_ais_init_r8, through the fused AIS call) is covered by the fixture
    tests at tile shape 8."""
    torch.manual_seed(D + K + 1)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)           # hidden width nodes x D in (192, 320]: the widths the 8-chain tiles exist for
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.03 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(5)
    eps = torch.randn(B, D, device=DEV, generator=g)
    ops = _ops.load()
    res = {}
    x0, _ = flow.native_sample(eps)
    try:
        for mode, shape in ((2, 8), (2, 8), (1, 8), (2, 16)):
            ops.set_option(_ops.OPT_R4_STREAM, mode)
            ops.set_option(_ops.OPT_TILE_SHAPE, shape)
            hmc = fa.HamiltonianMonteCarlo(3, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                           n_outer=2, L=4).to(DEV)
            pt = fa.create_point(x0.clone(), flow, target, with_grad=True)
            torch.manual_seed(7)
            out = hmc.transition(pt, 1, 0.3)
            res.setdefault((mode, shape), []).append((out.x.clone(), out.log_q.clone(), out.grad_log_q.clone(), hmc.epsilons.clone()))
    finally:
        ops.set_option(_ops.OPT_R4_STREAM, 2)
        ops.set_option(_ops.OPT_TILE_SHAPE, 0)
    for a, b in zip(res[(2, 8)][0], res[(2, 8)][1]):
        assert torch.equal(a, b)
    for other in ((1, 8), (2, 16)):
        xa, lqa, ga, ea = res[(2, 8)][0]
        xb, lqb, gb, eb = res[other][0]
        sc = max(1.0, float(xb.abs().max()))
        err = (xa - xb).abs().max(1).values / sc
        ok = err <= 1e-5
        assert int((~ok).sum()) <= max(1, B // 200), (other, int((~ok).sum()), float(err.max()))
        assert float((lqa[ok] - lqb[ok]).abs().max()) <= 2e-5 * max(1.0, float(lqb.abs().max()))
        gerr = (ga - gb).abs().max(1).values / max(1.0, float(gb.abs().max()))
        assert int(((gerr > 1e-4) & ok).sum()) <= max(1, B // 100), (other, int(((gerr > 1e-4) & ok).sum()), float(gerr.max()))
        if bool(ok.all()):
            assert torch.equal(ea, eb)
    # the fused and the unfused kernel are different roundings of the same function, not the same bits
    assert not torch.equal(res[(2, 8)][0][1], res[(1, 8)][0][1]) or B < 16


@pytest.mark.parametrize("D,K,nodes,B", [(32, 10, 10, 1024), (32, 10, 8, 37), (16, 3, 20, 257), (6, 3, 40, 70)])
def test_lds_prefetch_of_the_fused_four_chain_kernel_is_bit_identical_to_the_ring_alone(D, K, nodes, B):
    """FABHIP_OPT_R4_STREAM = 3 (opt-in: measured level with option 2): three items of every W x W stage of
    the fused 4-chain transition kernel reach the MFMAs through LDS (global_load_lds_dwordx4 issued during the short stages,
    ds_read_b128 when their turn comes) instead of the register ring.  Same tiles, same order of MFMAs: everything a transition
    returns is bit-identical to option 2, over two transitions from the same start, and repeated runs agree (a tile read before
    its copy has landed would show here)."""
    torch.manual_seed(D + K + 2)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.03 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device=DEV).manual_seed(5)
    x0, _ = flow.native_sample(torch.randn(B, D, device=DEV, generator=g))
    ops = _ops.load()
    res = {}
    try:
        for mode in (3, 2, 3, 3):
            ops.set_option(_ops.OPT_R4_STREAM, mode)
            ops.set_option(_ops.OPT_TILE_SHAPE, 4)
            hmc = fa.HamiltonianMonteCarlo(3, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                           n_outer=2, L=5).to(DEV)
            pt = fa.create_point(x0.clone(), flow, target, with_grad=True)
            torch.manual_seed(7)
            out = hmc.transition(pt, 1, 0.3)
            out = hmc.transition(out, 2, 0.6)
            res.setdefault(mode, []).append((out.x.clone(), out.log_q.clone(), out.grad_log_q.clone(), hmc.epsilons.clone()))
    finally:
        ops.set_option(_ops.OPT_R4_STREAM, 2)
        ops.set_option(_ops.OPT_TILE_SHAPE, 0)
    for other in (res[2][0], res[3][1], res[3][2]):
        for a, b in zip(res[3][0], other):
            assert torch.equal(a, b)


def _poisoned_noise(B, D, M, dev, rows, seed):
    """AIS noise of a run in which the chains `rows` die at "chain init" (NaN base noise: the compaction has rows to move) and two
    proposals of the last transition are NaN (rejected: a chain cannot die inside a transition)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    eps0 = torch.randn(B, D, device=dev, generator=g)
    noise_a = torch.randn(M, 1, B, D, device=dev, generator=g)
    noise_b = torch.empty(M, 1, B, device=dev).exponential_(1.0, generator=g)
    for r in rows:
        eps0[r, 0] = float("nan")
    for r in (5, B - 2):
        noise_a[M - 1, 0, r, 1] = float("nan")
    return eps0, noise_a, noise_b


@pytest.mark.parametrize("B,rows", [(1024, ()), (1024, (0, 3, 517, 1023)), (2048, (7, 1000, 2047)), (300, (299,)), (37, (0, 1, 2))])
def test_one_launch_tail_and_in_kernel_step_size_rule_are_bit_identical_to_the_separate_kernels(B, rows):
    """fabhip_ais_run with FABHIP_OPT_FUSED_TAIL / FABHIP_OPT_ADAPT_FOLD on (default: compaction + log_p - log_q + ESS / log Z in
    one launch per phase, k_tail_small; the step-size rule in the last workgroup of every transition kernel, hmc_adapt_last)
    against both off (k_valid_scan / k_compact_* / k_sub / k_ess_* and k_hmc_adapt as launches of their own): every output of
    the call, the counts, the statistics and every step size bit for bit - with chains dying at "chain init" (rows move in the
    compaction), on 4-chain (B <= 1152) and 8-chain tiles."""
    D, K, nodes, M = 32, 4, 320 // 32, 3
    torch.manual_seed(11)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    eps0, noise_a, noise_b = _poisoned_noise(B, D, M, DEV, rows, seed=B)
    outs = {}
    ops = _ops.load()
    try:
        for mode in (1, 0):
            ops.set_option(_ops.OPT_FUSED_TAIL, mode)
            ops.set_option(_ops.OPT_ADAPT_FOLD, mode)
            hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                           n_outer=1, L=3).to(DEV)
            ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=2.0,
                                               n_intermediate_distributions=M)
            res = []
            for _ in range(2):                                     # the second call starts from the first call's step sizes
                pt, log_w, n_valid, stats, base_x, base_lw = ais.run(B, eps0, noise_a, noise_b, want_base=True)
                n0, n1 = (int(v) for v in n_valid.cpu())
                res += [pt.x[:n1].clone(), pt.log_q[:n1].clone(), pt.log_p[:n1].clone(), pt.grad_log_q[:n1].clone(),
                        pt.grad_log_p[:n1].clone(), log_w[:n1].clone(), n_valid.clone(), stats[:6].clone(), base_x[:n0].clone(),
                        base_lw[:n0].clone(), hmc.epsilons.clone(), hmc.common_epsilon.clone()]
            outs[mode] = res
    finally:
        ops.set_option(_ops.OPT_FUSED_TAIL, 1)
        ops.set_option(_ops.OPT_ADAPT_FOLD, 1)
    n0, n1 = (int(v) for v in outs[1][6].cpu())
    assert n0 == B - len(rows) and n1 == n0
    for a, b in zip(outs[1], outs[0]):
        assert a.shape == b.shape and torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                                  b.view(torch.int32) if b.dtype == torch.float32 else b)
    assert not torch.equal(outs[1][-2], torch.full_like(outs[1][-2], 0.05))        # the rule ran: the step sizes moved


@pytest.mark.parametrize("B,metropolis", [(1024, False), (2048, False), (4096, False), (300, True)])
def test_noise_drawn_inside_the_op_equals_noise_drawn_by_the_caller(B, metropolis):
    """`AnnealedImportanceSampler.run` without noise tensors lets the op draw the transition noise itself - after the chain
    initialisation is enqueued (FABHIP_AIS_INIT, then FABHIP_AIS_CONTINUE | FABHIP_AIS_FINISH), from the default generator in
    the order the Python side would have used.  Same seed: the call equals the one-piece call on caller-drawn noise bit for
    bit (4-, 8- and 16-chain tiles, HMC with the in-kernel step-size rule and Metropolis)."""
    D, K, M = 32, 3, 3
    torch.manual_seed(21)
    flow = fa.RealNVP(D, K, 10).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    outs = []
    for inside in (True, False):
        if metropolis:
            op = fa.Metropolis(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, n_updates=2).to(DEV)
            n_inner = 2
        else:
            op = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                          n_outer=1, L=2).to(DEV)
            n_inner = 1
        ais = fa.AnnealedImportanceSampler(flow, target.log_prob, op, p_target=False, alpha=2.0, n_intermediate_distributions=M)
        torch.manual_seed(1000 + B)
        if inside:
            res = ais.run(B)
        else:
            eps0 = torch.randn(B, D, device=DEV)
            na = torch.randn(M, n_inner, B, D, device=DEV)
            nb = torch.rand(M, n_inner, B, device=DEV) if metropolis else torch.empty(M, n_inner, B, device=DEV).exponential_(1.0)
            res = ais.run(B, eps0, na, nb)
        pt, log_w, n_valid, stats = res[0], res[1], res[2], res[3]
        state = op.noise_scalings.clone() if metropolis else torch.cat([op.epsilons.flatten(), op.common_epsilon.flatten()])
        outs.append((pt.x.clone(), pt.log_q.clone(), pt.log_p.clone(), log_w.clone(), n_valid.clone(), stats[:6].clone(), state))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D,K,nodes,B", [(32, 10, 10, 1024), (32, 4, 8, 300), (6, 3, 40, 70), (16, 3, 16, 257), (2, 2, 64, 33),
                                         (32, 2, 4, 1152), (32, 3, 10, 4)])
def test_four_chain_flow_sample_matches_sixteen_chain_sample_and_the_oracle(D, K, nodes, B):
    """fabhip_flow_sample on 4-chain tiles (flow_sample_r4s: batches of <= 1152 chains, D <= 32, hidden width 128 .. 320; r4)
    against the 16-chain kernel (FABHIP_OPT_TILE_SHAPE = 16) on the same base noise - x and log q to fp32 rounding (the two
    differ in the summation order inside the products only) - against the density of its own samples, and against the CPU
    oracle's sampling direction."""
    from test_gpu_parity import seeded_flow, hip_flow_from_oracle
    nf = seeded_flow(D, K, nodes, 70 + D + K)
    hf = hip_flow_from_oracle(nf)
    torch.manual_seed(B)
    eps = torch.randn(B, D)
    ops = _ops.load()
    try:
        ops.set_option(_ops.OPT_TILE_SHAPE, 16)
        x16, lq16 = hf.native_sample(eps.to(DEV))
    finally:
        ops.set_option(_ops.OPT_TILE_SHAPE, 0)
    x4, lq4 = hf.native_sample(eps.to(DEV))
    xo, lqo = (t.detach() for t in nf.sample_eps(eps))
    assert max_rel_err(x4, x16) <= 5e-6 and max_rel_err(lq4, lq16) <= 5e-6
    assert max_rel_err(x4.cpu(), xo) <= 1e-5 and max_rel_err(lq4.cpu(), lqo) <= 1e-5
    assert max_rel_err(hf.log_prob(x4), lq4) <= 1e-5


@pytest.mark.parametrize("B", [1024, 2048])
def test_in_kernel_step_size_rule_is_race_free_over_many_calls(B):
    """The last-wave step-size rule (hmc_adapt_last: device-scope relaxed atomics + a ticket across the 8 XCDs' L2s) against the
    separate k_hmc_adapt launch over 120 consecutive AIS calls of the headline architecture (8 transitions each, step sizes
    carried from call to call): the step-size state after every call and a checksum of every call's log-weights are bit-identical.
    A lost or stale per-chain value anywhere in ~1000 launches would change a sum and with it every later step size."""
    D, K, M = 32, 10, 8
    torch.manual_seed(4)
    flow = fa.RealNVP(D, K, 10).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.03 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    ops = _ops.load()
    trace = {}
    try:
        for mode in (1, 0):
            ops.set_option(_ops.OPT_ADAPT_FOLD, mode)
            hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1,
                                           n_outer=1, L=2).to(DEV)
            ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=2.0,
                                               n_intermediate_distributions=M)
            torch.manual_seed(99)
            rows = []
            for _ in range(120):
                res = ais.run(B)
                rows.append(torch.cat([hmc.epsilons.flatten(), hmc.common_epsilon.flatten(), res[1].double().sum().float().reshape(1),
                                       res[3][:6]]).clone())
            trace[mode] = torch.stack(rows)
    finally:
        ops.set_option(_ops.OPT_ADAPT_FOLD, 1)
    a, b = trace[1].view(torch.int32), trace[0].view(torch.int32)
    bad = (a != b).any(dim=1).nonzero().flatten()
    assert bad.numel() == 0, f"first differing call: {int(bad[0])} of 120"
    assert not torch.equal(trace[1][0, :M], trace[1][-1, :M])                        # the step sizes did move
