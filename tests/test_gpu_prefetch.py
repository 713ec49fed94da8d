"""Repeated identical AIS calls (evaluation loops, the benchmark): from the second identical call on,
`AnnealedImportanceSampler.sample_and_log_weights` runs in two pieces of the fused call's own code (fabhip_ais_phase) and
enqueues the NEXT call's chain initialisation behind its device-to-host read.  Same kernels, same draws in the same order:
every call's result must be bit-identical to the one-op call's, and a changed parameter must drop the prefetched piece."""
import pytest
import torch

pytestmark = pytest.mark.gpu
fa = pytest.importorskip("fab_torch_amd")
DEV = "cuda"


def _sampler(D=32, K=3, nodes=10, M=4, eps=0.15, seed=0):
    torch.manual_seed(seed)
    flow = fa.make_wrapped_normflow_realnvp(D, K, nodes, act_norm=False).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.add_(0.01 * torch.randn_like(l3.weight))
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=eps, L=3).to(DEV)
    return flow, hmc, fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)


def _run(prefetch, n_calls, B, perturb_at=None, flat=False):
    flow, hmc, ais = _sampler()
    opt = fa.FlatAdam(flow.requires_grad_(True), lr=1e-3) if flat else None
    ais.prefetch = prefetch
    torch.manual_seed(11)
    outs = []
    for c in range(n_calls):
        if c == perturb_at:
            with torch.no_grad():
                if flat:      # FlatAdam's way: the flat image moves behind autograd's version counters, the image key is cleared
                    opt.theta.detach().mul_(1.01)
                    flow._packed_key = None
                else:
                    next(iter(flow.parameters())).mul_(1.01)      # a parameter moves between two calls (a training step)
        pt, lw = ais.sample_and_log_weights(B)
        info = ais.get_logging_info()
        outs.append((pt.x.clone(), pt.log_q.clone(), pt.log_p.clone(), lw.clone(), hmc.epsilons.clone(), hmc.common_epsilon.clone(),
                     info["ess_ais"], info["log_Z"], info["dist0_p_accept_0"]))
    return outs, ais


@pytest.mark.parametrize("B", [256, 1500, 2500])
def test_repeated_calls_with_prefetch_are_bit_identical_to_the_one_op_calls(B):
    a, _ = _run(False, 5, B)
    b, ais = _run(True, 5, B)
    assert "_pf_state" in ais.__dict__                                  # the sixth call's chain initialisation is already enqueued
    for c, (ra, rb) in enumerate(zip(a, b)):
        for i in range(6):
            assert torch.equal(ra[i], rb[i]), (c, i)
        assert ra[6:] == rb[6:], c


@pytest.mark.parametrize("flat", [False, True])
def test_a_changed_parameter_drops_the_prefetched_chain_initialisation(flat):
    a, _ = _run(False, 6, 512, perturb_at=3, flat=flat)
    b, _ = _run(True, 6, 512, perturb_at=3, flat=flat)
    assert not torch.equal(a[2][0], a[3][0])
    for c, (ra, rb) in enumerate(zip(a, b)):
        for i in range(6):
            assert torch.equal(ra[i], rb[i]), (c, i)
    if flat:
        return
    # ... and so does another batch size or explicit noise
    flow, hmc, ais = _sampler()
    for B in (256, 256, 256, 128, 128):
        ais.sample_and_log_weights(B)
    pt, lw = ais.sample_and_log_weights(64, eps0=torch.randn(64, 32, device=DEV))
    assert pt.x.shape[0] == 64
