"""The PyTorch-ROCm custom-op layer (torch.ops.fabhip.*, csrc/torch_ops.cpp) against the raw C ABI of libfabhip.so
driven through ctypes (fab_torch_amd/_lib.py) on the same inputs: results must be BIT-identical (the ops only
marshal tensors into the C structs).  Plus: dispatcher behaviour (CUDA/HIP key only), stream handling, the
registered autograd of fabhip::realnvp_logprob_tape, torch.library.opcheck."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from fab_torch_amd import _lib, _ops          # noqa: E402
from oracle import flow as oflow              # noqa: E402

DEV = "cuda"


def _flow(D, K, nodes, seed):
    torch.manual_seed(seed)
    nf = oflow.make_realnvp(D, K, nodes)
    oflow.randomize_last_layers(nf, 0.05, seed + 1)
    f = fa.RealNVP(D, K, nodes)
    f._nf_model.load_state_dict(nf.state_dict())
    return f.to(DEV).requires_grad_(False)


def _cabi_flow(flow):
    packed, D, K, W = flow.native()
    return _lib.Flow(D, K, W, packed.data_ptr()), packed


def _sync_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_ops_are_registered_for_the_hip_key_only():
    ops = _ops.load()
    assert ops.abi_version() == _ops.ABI_VERSION == _lib.load().fabhip_version()
    with pytest.raises(NotImplementedError, match="CPU"):
        ops.ess_logz(torch.randn(8), None, 8.0)
    with pytest.raises(NotImplementedError, match="CPU"):
        ops.manywell_logp_grad(torch.randn(4, 6), -0.5, -6.0, 1.0, 0.0)
    with pytest.raises(RuntimeError, match="fabhip"):           # C-ABI error code -> c10::Error
        ops.realnvp_sample(torch.zeros(7, device=DEV), 6, 2, 30, torch.randn(4, 6, device=DEV))


@pytest.mark.parametrize("D,K,nodes,B", [(6, 3, 5, 50), (32, 10, 10, 77)])
def test_realnvp_ops_equal_the_c_abi_bit_for_bit(D, K, nodes, B):
    lib, ops = _lib.load(), _ops.load()
    flow = _flow(D, K, nodes, 3)
    f, packed = _cabi_flow(flow)
    x = torch.randn(B, D, device=DEV)
    lq_c = torch.empty(B, device=DEV); g_c = torch.empty(B, D, device=DEV)
    _lib.check(lib.fabhip_flow_log_prob(C.byref(f), _lib.ptr(x), _lib.ptr(lq_c), _lib.ptr(g_c), B, _sync_stream()))
    lq_o, g_o = ops.realnvp_logprob_grad(packed, D, K, flow.width, x, True)
    assert torch.equal(lq_o, lq_c) and torch.equal(g_o, g_c)
    eps = torch.randn(B, D, device=DEV)
    xs_c = torch.empty(B, D, device=DEV); lqs_c = torch.empty(B, device=DEV)
    _lib.check(lib.fabhip_flow_sample(C.byref(f), _lib.ptr(eps), _lib.ptr(xs_c), _lib.ptr(lqs_c), B, _sync_stream()))
    xs_o, lqs_o = ops.realnvp_sample(packed, D, K, flow.width, eps)
    assert torch.equal(xs_o, xs_c) and torch.equal(lqs_o, lqs_c)
    # on a side stream: the op enqueues on torch's CURRENT stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lq_s, _ = ops.realnvp_logprob_grad(packed, D, K, flow.width, x, False)
    side.synchronize()
    assert torch.equal(lq_s, lq_c)


def test_ais_run_op_equals_the_c_abi_bit_for_bit():
    lib, ops = _lib.load(), _ops.load()
    D, K, nodes, M, B, L = 6, 3, 5, 4, 100, 5
    flow = _flow(D, K, nodes, 5)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=L).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
    torch.manual_seed(0)
    eps0 = torch.randn(B, D, device=DEV); na = torch.randn(M, 1, B, D, device=DEV)
    nb = torch.empty(M, 1, B, device=DEV).exponential_()
    eps_in, ceps_in = hmc.epsilons.clone(), hmc.common_epsilon.clone()
    # (1) the op, through the product class
    pt, lw, n_valid, stats, _, _ = ais.run(B, eps0, na, nb)
    eps_op, ceps_op = hmc.epsilons.clone(), hmc.common_epsilon.clone()
    # (2) the raw C ABI through ctypes on the same inputs / the same initial step sizes
    eps_c, ceps_c = eps_in.clone(), ceps_in.clone()
    f, packed = _cabi_flow(flow)
    a = _lib.AisArgs()
    a.flow = f
    kind, prm, _, _ = target.native_target()
    a.target = _lib.Target(kind, D, prm[0], prm[1], prm[2], prm[3], 0, None, None)
    betas = (C.c_double * (M + 2))(*[float(b) for b in ais.B_space])
    a.B, a.M, a.betas, a.alpha, a.p_target, a.transition = B, M, betas, 2.0, 0, _lib.TRANSITION_HMC
    a.eps0, a.noise_a, a.noise_b = eps0.data_ptr(), na.data_ptr(), nb.data_ptr()
    a.step_state, a.common_epsilon, a.mass = eps_c.data_ptr(), ceps_c.data_ptr(), hmc.mass_vector.data_ptr()
    a.n_inner, a.L, a.max_grad, a.target_p_accept, a.tune = 1, L, hmc.max_grad, hmc.target_p_accept, 1
    f32 = dict(dtype=torch.float32, device=DEV)
    x, lq, lp, lwc = torch.empty(B, D, **f32), torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32)
    gq, gp = torch.empty(B, D, **f32), torch.empty(B, D, **f32)
    nv, st = torch.zeros(2, dtype=torch.int32, device=DEV), torch.zeros(16, **f32)
    a.point = _lib.Point(x.data_ptr(), lq.data_ptr(), lp.data_ptr(), gq.data_ptr(), gp.data_ptr())
    a.log_w, a.n_valid, a.stats = lwc.data_ptr(), nv.data_ptr(), st.data_ptr()
    nbytes = lib.fabhip_ais_workspace_bytes(B, D, 1)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=DEV)
    a.workspace, a.workspace_bytes = (ws.data_ptr() + 255) // 256 * 256, nbytes
    _lib.check(lib.fabhip_ais_run(C.byref(a), _sync_stream()), "ais_run")
    torch.cuda.synchronize()
    assert torch.equal(nv, n_valid) and int(nv[1]) == B
    for got, ref in ((pt.x, x), (pt.log_q, lq), (pt.log_p, lp), (pt.grad_log_q, gq), (pt.grad_log_p, gp), (lw, lwc),
                     (stats[:6], st[:6]), (eps_op, eps_c), (ceps_op, ceps_c)):
        assert torch.equal(got, ref)
    # the op also reachable directly, by name
    out = torch.ops.fabhip.ais_run(*flow.native(), *target.native_target(), [float(b) for b in ais.B_space], 2.0, False,
                                   _ops.TRANSITION_HMC, eps0, na, nb, eps_in.clone(), ceps_in.clone(), hmc.mass_vector,
                                   1, L, float(hmc.max_grad), float(hmc.target_p_accept), True, None, None, None, None,
                                   False)
    assert torch.equal(out[0], x) and torch.equal(out[5], lwc)


def test_logprob_tape_autograd_matches_explicit_param_grad_and_opcheck():
    ops = _ops.load()
    D, K, nodes, B = 6, 3, 5, 40
    flow = _flow(D, K, nodes, 9).requires_grad_(True)
    x = torch.randn(B, D, device=DEV, requires_grad=True)
    coef = torch.randn(B, device=DEV)
    lq = flow.log_prob(x)                               # torch.ops.fabhip.realnvp_logprob_tape + registered autograd
    assert lq.requires_grad
    (lq * coef).sum().backward()
    with torch.no_grad():
        lq2, handle, gx = flow.log_prob_with_tape(x, want_grad_x=True)
        flat = flow.param_grad_flat(handle, coef)
    assert torch.equal(lq.detach(), lq2) and torch.equal(x.grad, coef[:, None] * gx)
    views = flow._grad_views(flat)
    for p, v in zip(flow._grad_tensors(), views):
        assert torch.equal(p.grad, v), "autograd of the op != fabhip::realnvp_param_grad"
    # registration sanity (schema, fake tensor where registered, autograd registration)
    packed, _, _, W = flow.native(need_inverse=False)
    torch.library.opcheck(torch.ops.fabhip.realnvp_logprob_grad.default, (packed, D, K, W, x.detach(), True),
                          test_utils=("test_schema", "test_faketensor"))
    theta = torch.cat([p.detach().reshape(-1) for p in flow._grad_tensors()]).requires_grad_(True)
    torch.library.opcheck(torch.ops.fabhip.realnvp_logprob_tape.default,
                          (theta, x.detach(), packed, [p.detach() for p in flow._param_list()], D, K, W, False),
                          test_utils=("test_schema", "test_autograd_registration"))


def test_target_and_resample_ops_equal_the_c_abi():
    lib, ops = _lib.load(), _ops.load()
    x = torch.randn(333, 32, device=DEV) * 1.5
    lp, g = ops.manywell_logp_grad(x, -0.5, -6.0, 1.0, 0.0)
    t = _lib.Target(_lib.TARGET_MANYWELL, 32, -0.5, -6.0, 1.0, 0.0, 0, None, None)
    lp_c, g_c = torch.empty(333, device=DEV), torch.empty(333, 32, device=DEV)
    _lib.check(lib.fabhip_target_log_prob(C.byref(t), _lib.ptr(x), _lib.ptr(lp_c), _lib.ptr(g_c), 333, _sync_stream()))
    assert torch.equal(lp, lp_c) and torch.equal(g, g_c)
    lw = torch.randn(100_003, device=DEV) * 3
    idx = ops.resample_systematic(lw, 0.25, 100_003)
    idx_c = torch.empty(100_003, dtype=torch.int64, device=DEV)
    nb = lib.fabhip_resample_workspace_bytes(100_003)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=DEV)
    _lib.check(lib.fabhip_resample_systematic(_lib.ptr(lw), 100_003, 0.25, 100_003, _lib.ptr(idx_c),
                                              C.c_void_p((ws.data_ptr() + 255) // 256 * 256), nb, _sync_stream()))
    assert torch.equal(idx, idx_c)
    st = ops.ess_logz(lw, None, float(lw.numel()))
    st_c = torch.empty(3, device=DEV)
    nb = lib.fabhip_ess_workspace_bytes(lw.numel())
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=DEV)
    _lib.check(lib.fabhip_ess_logz(_lib.ptr(lw), lw.numel(), None, float(lw.numel()), _lib.ptr(st_c),
                                   C.c_void_p((ws.data_ptr() + 255) // 256 * 256), nb, _sync_stream()))
    assert torch.equal(st, st_c)


def test_fused_ais_call_is_capturable_in_a_hip_graph():
    """The op layer enqueues on torch's current stream, takes every buffer from the caching allocator and never
    synchronises: a whole AIS call (device RNG + fabhip::ais_run, all M transitions, compaction, ESS) records into ONE
    HIP graph through torch.cuda.graph and replays with fresh noise (tools/try_graph.py times eager vs replay: equal -
    the call is not launch-bound)."""
    D, K, M, B = 6, 3, 4, 256
    torch.manual_seed(0)
    flow = fa.RealNVP(D, K, 8).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            ais.run(B)
    torch.cuda.current_stream().wait_stream(s)
    eps_before = hmc.epsilons.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        point, log_w, n_valid, stats, _, _ = ais.run(B)
    g.replay(); torch.cuda.synchronize()
    lw1, x1 = log_w.clone(), point.x.clone()
    g.replay(); torch.cuda.synchronize()
    assert torch.isfinite(log_w).all() and int(n_valid[1]) == B
    assert not torch.equal(lw1, log_w) and not torch.equal(x1, point.x)          # the device RNG advances per replay
    assert not torch.equal(eps_before, hmc.epsilons)                              # step sizes keep adapting on the device
    ref = fa.effective_sample_size(log_w)
    assert 0 < float(ref) <= 1


def test_reference_helper_methods_of_the_operators_sampler_and_target(tmp_path):
    """Small public methods of the reference a caller may use next to the hot path: TransitionOperator.
    (grad_)intermediate_target_log_prob (transition_operators/base.py:37-54), AnnealedImportanceSampler.perform_transition
    (ais.py:90-105), HamiltonianMonteCarlo.save_model / load_model (hmc.py:204-222), ManyWellEnergy.log_prob_2D / energy /
    force / sample_first_dimension (many_well.py:92-94, double_well.py:19-28,60-82)."""
    D, M, B = 6, 3, 64
    torch.manual_seed(0)
    flow = fa.RealNVP(D, 2, 6).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
    x, _ = flow.sample_and_log_prob((B,))
    pt = hmc.create_new_point(x)
    beta, alpha = 0.3, 2.0
    ref = ((1 - beta) + beta * (1 - alpha)) * pt.log_q + beta * alpha * pt.log_p
    assert torch.allclose(hmc.intermediate_target_log_prob(pt, beta), ref, rtol=1e-5, atol=1e-5)
    gref = ((1 - beta) + beta * (1 - alpha)) * pt.grad_log_q + 2 * beta * pt.grad_log_p          # the reference's 2 beta
    assert torch.allclose(hmc.grad_intermediate_target_log_prob(pt, beta), gref, rtol=1e-5, atol=1e-5)
    hmc.p_target = True
    assert torch.allclose(hmc.intermediate_target_log_prob(pt, beta), (1 - beta) * pt.log_q + beta * pt.log_p, rtol=1e-5, atol=1e-5)
    hmc.p_target = False
    assert torch.equal(fa.get_intermediate_log_prob(pt, beta, alpha, False), hmc.intermediate_target_log_prob(pt, beta))
    assert torch.equal(fa.get_grad_intermediate_log_prob(pt, beta, alpha, False), hmc.grad_intermediate_target_log_prob(pt, beta))
    # perform_transition = transition + log-weight increment, input log_w untouched
    lw0 = torch.zeros(B, device=DEV)
    torch.manual_seed(5)
    p1 = fa.Point(pt.x.clone(), pt.log_q.clone(), pt.log_p.clone(), pt.grad_log_q.clone(), pt.grad_log_p.clone())
    p1, lw1 = ais.perform_transition(p1, lw0, 1)
    torch.manual_seed(5)
    p2 = fa.Point(pt.x.clone(), pt.log_q.clone(), pt.log_p.clone(), pt.grad_log_q.clone(), pt.grad_log_p.clone())
    hmc.epsilons.copy_(torch.ones_like(hmc.epsilons) * 0.2 * 0.9); hmc.common_epsilon.fill_(0.02)
    assert torch.equal(lw0, torch.zeros(B, device=DEV)) and torch.isfinite(lw1).all() and not torch.equal(lw1, lw0)
    b1, b2 = float(ais.B_space[1]), float(ais.B_space[2])
    inc = hmc.intermediate_target_log_prob(p1, b2) - hmc.intermediate_target_log_prob(p1, b1)
    assert torch.allclose(lw1, inc, rtol=1e-4, atol=1e-4)
    # HMC save / load
    hmc.save_model(tmp_path, epoch=3)
    other = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=1.0, L=3).to(DEV)
    other.load_model(tmp_path, epoch=3, device=DEV)
    assert torch.equal(other.epsilons, hmc.epsilons) and torch.equal(other.common_epsilon, hmc.common_epsilon)
    # the 2-D helpers of the target
    x2 = torch.randn(32, 2, device=DEV)
    full = torch.zeros(32, D, device=DEV); full[:, :2] = x2
    rest = target.log_prob(torch.zeros(1, D, device=DEV))[0] * (D // 2 - 1) / (D // 2)        # the other wells at the origin
    assert torch.allclose(target.log_prob_2D(x2), target.log_prob(full) - rest, rtol=1e-5, atol=1e-5)
    assert target.energy(x2).shape == (32, 1) and torch.allclose(target.energy(x2)[:, 0], -target.log_prob_2D(x2))
    f = target.force(x2)
    assert torch.allclose(f[:, 0], 0.5 + 12 * x2[:, 0] - 4 * x2[:, 0] ** 3, rtol=1e-4, atol=1e-4) and torch.allclose(f[:, 1], -x2[:, 1])
    s = target.sample_first_dimension((4096,))
    assert s.shape == (4096,) and 0.6 < float((s > 0).float().mean()) < 0.97            # the deep well is at +1.7


def test_experimental_losses_are_refused_like_the_reference_and_the_baseline_losses_differentiate():
    """core.py:13-15,50-51: FABModel refuses the three experimental loss types with the reference's message (their bodies are
    not carried); the two reparameterised baseline losses the constructor accepts differentiate through the HIP ops."""
    D, M, B = 6, 2, 64
    torch.manual_seed(0)
    flow = fa.RealNVP(D, 2, 6).to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=3).to(DEV)
    for lt in ("flow_alpha_2_div", "flow_alpha_2_div_unbiased", "fab_ub_alpha_2_div"):
        with pytest.raises(Exception, match="experiment loss"):
            fa.FABModel(flow, target, M, alpha=2.0, transition_operator=hmc, loss_type=lt)
    model = fa.FABModel(flow, target, M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    assert not hasattr(model, "flow_alpha_2_div") and not hasattr(model, "fab_ub_alpha_div_loss")
    for fn in (model.flow_alpha_2_div_nis, model.flow_reverse_kl):
        for p in flow.parameters():
            p.grad = None
        loss = fn(B)
        assert loss.dim() == 0 and torch.isfinite(loss)
        loss.backward()
        assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in flow.parameters()), fn.__name__


@pytest.mark.parametrize("B", [100, 33, 1000])
def test_four_chain_tiles_never_read_uninitialised_scratch(B):
    """with 4 chains per workgroup, k_hmc_adapt sums all 16 rows of every 16-row block, so every row up
    to ceil16(B) has to be written by SOME workgroup.  The C ABI's caller-owned workspace is pre-filled with NaN bytes:
    acceptance logging and the adapted step sizes must be finite and identical to a run on a zeroed workspace."""
    lib = _lib.load()
    D, K, nodes, M, L = 6, 3, 5, 3, 3
    flow = _flow(D, K, nodes, 5)
    target = fa.ManyWellEnergy(D)
    f, packed = _cabi_flow(flow)
    kind, prm, _, _ = target.native_target()
    tgt = _lib.Target(kind, D, prm[0], prm[1], prm[2], prm[3], 0, None, None)
    g = torch.Generator(device=DEV).manual_seed(B)
    x0, _ = flow.native_sample(torch.randn(B, D, device=DEV, generator=g))
    noise_p = torch.randn(1, B, D, device=DEV, generator=g)
    noise_e = torch.empty(1, B, device=DEV).exponential_(generator=g)
    mass = torch.ones(D, device=DEV)
    res = []
    with _ops.option(_ops.OPT_TILE_SHAPE, 4):
        for fill in (0xFF, 0x00):
            pt = fa.create_point(x0.clone(), flow, target, with_grad=True)
            lw = torch.zeros(B, device=DEV)
            eps, ceps = torch.full((1,), 0.1, device=DEV), torch.full((1,), 0.01, device=DEV)
            pacc, dist = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
            nbytes = lib.fabhip_hmc_workspace_bytes(B, D, 1)
            ws = torch.full((nbytes + 256,), fill, dtype=torch.uint8, device=DEV)
            a = _lib.HmcArgs()
            a.flow, a.target = f, tgt
            a.point = _lib.Point(pt.x.data_ptr(), pt.log_q.data_ptr(), pt.log_p.data_ptr(), pt.grad_log_q.data_ptr(),
                                 pt.grad_log_p.data_ptr())
            a.B, a.n_valid = B, None
            lib.fabhip_anneal_coefs(0.3, 2.0, 0, C.byref(a.cur)); lib.fabhip_anneal_coefs(0.6, 2.0, 0, C.byref(a.next))
            a.log_w, a.noise_p, a.noise_e = lw.data_ptr(), noise_p.data_ptr(), noise_e.data_ptr()
            a.epsilons, a.common_epsilon, a.mass = eps.data_ptr(), ceps.data_ptr(), mass.data_ptr()
            a.n_outer, a.L, a.max_grad, a.target_p_accept, a.tune = 1, L, 1e3, 0.65, 1
            a.p_accept, a.avg_distance = pacc.data_ptr(), dist.data_ptr()
            a.workspace, a.workspace_bytes = (ws.data_ptr() + 255) // 256 * 256, nbytes
            _lib.check(lib.fabhip_hmc_transition(C.byref(a), _sync_stream()), "hmc_transition")
            torch.cuda.synchronize()
            res.append((pacc.clone(), dist.clone(), eps.clone(), ceps.clone(), lw.clone()))
    for t in res[0]:
        assert torch.isfinite(t).all()
    for got, ref in zip(res[0], res[1]):
        assert torch.equal(got, ref)
    assert 0.0 < float(res[0][0]) <= 1.0


def test_ops_reject_tensors_of_the_wrong_size_or_device():
    """every tensor whose pointer reaches a kernel is checked (element count, device) in the op layer -
    a wrong length is an error, never an out-of-bounds access; an HMC AIS run without common_epsilon / mass is refused."""
    ops = _ops.load()
    D, K, nodes, B, M = 6, 2, 5, 32, 2
    flow = _flow(D, K, nodes, 1)
    target = fa.ManyWellEnergy(D)
    fl, tg = flow.native(), target.native_target()
    x = torch.randn(B, D, device=DEV)
    lq, lp, gq, gp = ops.create_point(*fl, *tg, x, True)
    noise_p, noise_e = torch.randn(1, B, D, device=DEV), torch.rand(1, B, device=DEV)
    eps, ceps, mass = torch.full((1,), 0.1, device=DEV), torch.full((1,), 0.01, device=DEV), torch.ones(D, device=DEV)

    def hmc(**kw):
        a = dict(x=x.clone(), lq=lq.clone(), lp=lp.clone(), gq=gq.clone(), gp=gp.clone(), lw=torch.zeros(B, device=DEV),
                 noise_p=noise_p, noise_e=noise_e, eps=eps.clone(), ceps=ceps.clone(), mass=mass)
        a.update(kw)
        ops.hmc_transition(*fl, *tg, a["x"], a["lq"], a["lp"], a["gq"], a["gp"], a["lw"], 0.3, 0.6, 2.0, False,
                           a["noise_p"], a["noise_e"], a["eps"], a["ceps"], a["mass"], 3, 1e3, 0.65, True, None, None)

    hmc()                                                   # the well-formed call passes
    for bad in (dict(lw=torch.zeros(B - 1, device=DEV)), dict(lq=lq[:-1].clone()), dict(gq=gq[:, :-1].contiguous()),
                dict(mass=torch.ones(D + 1, device=DEV)), dict(ceps=torch.zeros(0, device=DEV)),
                dict(noise_e=noise_e[:, :-1].contiguous()), dict(mass=torch.ones(D))):
        with pytest.raises((RuntimeError, NotImplementedError), match="fabhip|CPU"):
            hmc(**bad)
    betas = [0.0, 0.3, 0.6, 1.0]
    eps0 = torch.randn(B, D, device=DEV)
    na, nb = torch.randn(M, 1, B, D, device=DEV), torch.rand(M, 1, B, device=DEV)
    step = torch.full((M, 1), 0.1, device=DEV)
    with pytest.raises(RuntimeError, match="common_epsilon and the mass"):
        ops.ais_run(*fl, *tg, betas, 2.0, False, _ops.TRANSITION_HMC, eps0, na, nb, step, None, None, 1, 3, 1e3, 0.65, True,
                    None, None, None, None, False)
    with pytest.raises(RuntimeError, match="mass"):
        ops.ais_run(*fl, *tg, betas, 2.0, False, _ops.TRANSITION_HMC, eps0, na, nb, step, ceps, torch.ones(D - 1, device=DEV),
                    1, 3, 1e3, 0.65, True, None, None, None, None, False)
    with pytest.raises(RuntimeError, match="w2"):           # a weight tensor of the wrong shape never reaches the pack kernel
        prm = [p.detach() for p in flow._param_list()]
        prm[2] = prm[2][:, :-1].contiguous()
        ops.realnvp_pack(prm, D, K, flow.width, True, torch.empty_like(fl[0]))


def test_option_table_is_read_without_the_environment():
    ops, lib = _ops.load(), _lib.load()
    assert ops.get_option(_ops.OPT_TILE_SHAPE) == 0 and ops.get_option(_ops.OPT_SCAN_VARIANT) == 3
    with _ops.option(_ops.OPT_TILE_SHAPE, 16):
        assert lib.fabhip_get_option(_ops.OPT_TILE_SHAPE) == 16
    assert ops.get_option(_ops.OPT_TILE_SHAPE) == 0
    assert lib.fabhip_set_option(99, 1) < 0
    with pytest.raises(RuntimeError, match="unknown option"):
        ops.set_option(99, 1)


def test_packed_image_follows_every_kind_of_parameter_update():
    """flow.native() decides from (storage address, version) of the REGISTERED parameter objects whether the packed image is
    current (one integer through the dispatcher per call).  Every way the values can change must be seen: in-place updates
    (optimizer steps), load_state_dict (also with assign=True), `param.data = ...`, .to() round trips, a re-assigned Parameter
    object of any layer (round 5: every registered object is compared by identity per call)."""
    torch.manual_seed(3)
    D, K = 6, 3
    flow = fa.RealNVP(D, K, 8).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():                # (the builder zero-initialises the conditioners' last layers)
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(40, D, device=DEV)

    def lq():
        return flow.log_prob(x).clone()

    def fresh():                                   # the same parameters through a flow that never cached anything
        f2 = fa.RealNVP(D, K, 8).to(DEV).requires_grad_(False)
        f2._nf_model.load_state_dict(flow._nf_model.state_dict())
        return f2.log_prob(x).clone()

    base = lq()
    assert torch.equal(base, fresh())
    mid = list(flow._layers())[1]                  # a layer none of the probes looks at
    with torch.no_grad():
        mid[1].weight.add_(0.1)                    # in place
    a = lq()
    assert not torch.equal(a, base) and torch.equal(a, fresh())
    mid[1].weight.data = mid[1].weight.data * 0.5  # .data assignment: new storage, same Parameter object
    b = lq()
    assert not torch.equal(b, a) and torch.equal(b, fresh())
    sd = {k: v.clone() * 1.01 for k, v in flow._nf_model.state_dict().items()}
    flow._nf_model.load_state_dict(sd)
    c = lq()
    assert not torch.equal(c, b) and torch.equal(c, fresh())
    flow.cpu()
    flow.to(DEV)                                    # .to(): _apply invalidates the registration
    assert torch.equal(lq(), c)
    first = list(flow._layers())[0]
    first[0].weight = torch.nn.Parameter(first[0].weight.detach() * 0.9, requires_grad=False)   # probed object replaced
    d = lq()
    assert not torch.equal(d, c) and torch.equal(d, fresh())
    mid = list(flow._layers())[1]
    mid[2].bias = torch.nn.Parameter(mid[2].bias.detach() + 0.2, requires_grad=False)           # a middle layer's object replaced:
    e = lq()                                                                                    # seen WITHOUT invalidate_native()
    assert not torch.equal(e, d) and torch.equal(e, fresh())                                    # (round 5: every object is probed)
    sd2 = {k: (v.clone() * 0.97) for k, v in flow._nf_model.state_dict().items()}
    flow._nf_model.load_state_dict(sd2, assign=True)                                             # assign=True swaps the objects
    f = lq()
    assert not torch.equal(f, e) and torch.equal(f, fresh())
    mid = list(flow._layers())[1]
    mid[3].log_S = torch.nn.Parameter(mid[3].log_S.detach() + 0.05, requires_grad=False)        # InvertibleAffine parameter
    g2 = lq()
    assert not torch.equal(g2, f) and torch.equal(g2, fresh())


def test_deep_copied_flow_registers_its_own_parameter_set():
    """A copy.deepcopy of a flow carries the source's cached state in __dict__ (the handle of the op layer's parameter-set
    registry among it); it must register its OWN tensors, follow its own updates, and its deletion must not release the
    source's entry."""
    import copy
    import gc
    torch.manual_seed(5)
    D, K = 6, 2
    flow = fa.RealNVP(D, K, 8).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(30, D, device=DEV)
    a0 = flow.log_prob(x).clone()
    twin = copy.deepcopy(flow)
    assert torch.equal(twin.log_prob(x), a0)
    with torch.no_grad():
        list(twin._layers())[0][1].weight.mul_(1.5)
    b = twin.log_prob(x).clone()
    assert not torch.equal(b, a0)
    assert torch.equal(flow.log_prob(x), a0)                  # the source did not see the copy's update ...
    with torch.no_grad():
        list(flow._layers())[1][0].weight.add_(0.05)
    a1 = flow.log_prob(x).clone()
    assert not torch.equal(a1, a0) and torch.equal(twin.log_prob(x), b)      # ... nor the copy the source's
    del twin
    gc.collect()
    with torch.no_grad():
        list(flow._layers())[1][0].weight.add_(0.05)
    assert not torch.equal(flow.log_prob(x), a1)              # the source's registration survived the copy's deletion


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the source tensor on a non-current device)")
def test_small_host_read_waits_for_a_copy_from_a_non_current_device():
    """`_ops._read_small` recorded its event on the CURRENT device's stream while the copy ran on the stream of
    the tensor's device - with the flow on cuda:1 and cuda:0 current the event completed at once and the pinned buffer was read
    before the copy landed.  Queue work on cuda:1 in front of the source, read it while cuda:0 is current."""
    from fab_torch_amd import _ops
    d1 = torch.device("cuda", 1)
    with torch.cuda.device(0):
        for rep in range(20):
            a = torch.randn(4096, 4096, device=d1)
            t = (a @ a).sum().reshape(1).expand(18).contiguous() * 0 + float(rep)      # the value exists only after the GEMM
            h = _ops._read_small(t)
            assert float(h[0]) == float(rep) and float(h[17]) == float(rep)
