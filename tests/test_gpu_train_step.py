"""GPU tests of the training half of a FAB iteration as built in round 6: the replay-buffer minibatch step as ONE op
(`fabhip::buffer_train_step`: fab/train_with_prioritised_buffer.py:158-185 + prioritised_replay_buffer.py:117-131), the buffer's
`add` and its row selection as one op each, the tape forward on 8-chain tiles and the tile GEMM of the parameter gradients
against the kernels they replace."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _flow(D, K, nodes, dev, act_norm=False, seed=0):
    import fab_torch_amd as fa
    torch.manual_seed(seed)
    flow = fa.make_wrapped_normflow_realnvp(D, K, nodes, act_norm=act_norm).to(dev)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.add_(0.01 * torch.randn_like(l3.weight))
            l3.bias.add_(0.01 * torch.randn_like(l3.bias))
    return flow


@pytest.mark.parametrize("D,K,nodes,B,an", [(32, 4, 10, 2048, False), (6, 8, 40, 1000, False), (2, 4, 40, 77, False),
                                            (60, 3, 4, 333, False), (33, 2, 8, 130, True), (12, 3, 20, 4096, False),
                                            (32, 2, 6, 5, False)])
def test_tile_gemm_parameter_gradients_equal_the_block_kernel(D, K, nodes, B, an):
    """k_pgrad_tiles (default) against the round-1 64 x 64 block kernel (FABHIP_PGRAD=0) on the same tape: every parameter
    tensor to 2e-5 of the image's largest entry (two fp32 summation orders of B terms), bitwise reproducible."""
    from fab_torch_amd import _ops
    dev = torch.device("cuda", 0)
    flow = _flow(D, K, nodes, dev, an)
    x = torch.randn(B, D, device=dev)
    coef = torch.randn(B, device=dev) / B
    lq, tape = flow.log_prob_with_tape(x)
    assert bool(torch.isfinite(lq).all())
    res = {}
    for mode in (0, 1):
        with _ops.option(_ops.OPT_PGRAD, mode):
            res[mode] = [flow.param_grad_flat(tape, coef).clone() for _ in range(3)]
    a, b = res[0][0], res[1][0]
    assert all(torch.equal(b, r) for r in res[1])
    assert bool(torch.isfinite(b).all())
    assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-7


@pytest.mark.parametrize("D,K,nodes,B", [(32, 3, 8, 100), (32, 10, 10, 515), (6, 8, 40, 1000), (12, 3, 20, 37), (32, 2, 10, 8)])
def test_eight_chain_tape_forward_matches_the_sixteen_chain_kernel(D, K, nodes, B):
    """fabhip_flow_log_prob_tape on 8-chain stream tiles (default where the flow has that image) against the 16-chain kernel
    (FABHIP_TAPE_TILES=16): log q to 1e-5; the parameter gradients of each tape against the float64 oracle are covered by
    test_gpu_parity.py::test_flow_parameter_gradients_vs_oracle_autograd (whichever kernel the shape selects); here the two
    kernels' gradients agree wherever no hidden unit sits within rounding distance of its ReLU kink (rows whose ReLU decisions
    agree in both tapes)."""
    from fab_torch_amd import _ops
    dev = torch.device("cuda", 0)
    ops = _ops.load()
    flow = _flow(D, K, nodes, dev)
    x = torch.randn(B, D, device=dev)
    lay = [int(v) for v in ops.flow_tape_layout(D, K, D * nodes, B)]
    Bp, wz, w1, wh, wp, we, wb, oZA, oGZ, oZ1, oH1, oH2, oDP, oE2, oE1, stride, oTB, total = lay
    out = {}
    for mode in (16, 0):
        with _ops.option(_ops.OPT_TAPE_TILES, mode):
            lq, tape, gx = flow.log_prob_with_tape(x, want_grad_x=True)
            lq2, tape2, _ = flow.log_prob_with_tape(x, want_grad_x=True)
            assert torch.equal(lq, lq2)
            out[mode] = (lq.clone(), gx.clone(), tape[0][:total].clone())
    a, b = out[16], out[0]
    assert float(((a[0] - b[0]).abs() / a[0].abs().clamp(min=1.0)).max()) < 1e-5
    # rows with identical ReLU decisions in every layer (H1 / H2 > 0 patterns): their tape rows and input gradients agree
    same = torch.ones(B, dtype=torch.bool, device=dev)
    for k in range(K):
        for o in (oH1, oH2):
            ha = a[2][k * stride + o: k * stride + o + Bp * wh].view(Bp, wh)[:B, :D * nodes]
            hb = b[2][k * stride + o: k * stride + o + Bp * wh].view(Bp, wh)[:B, :D * nodes]
            same &= ((ha > 0) == (hb > 0)).all(dim=1)
    assert int(same.sum()) >= (B * 3) // 4
    ga, gb = a[1][same], b[1][same]
    assert float((ga - gb).abs().max()) <= 2e-4 * float(ga.abs().max()) + 1e-6
    for k in range(K):
        for o, w in ((oZA, wz), (oGZ, wz), (oH1, wh), (oH2, wh), (oDP, wp), (oE2, we), (oE1, we)):
            ta = a[2][k * stride + o: k * stride + o + Bp * w].view(Bp, w)[:B][same]
            tb = b[2][k * stride + o: k * stride + o + Bp * w].view(Bp, w)[:B][same]
            assert float((ta - tb).abs().max()) <= 2e-4 * float(ta.abs().max()) + 1e-6, (k, o)


def test_tape_forward_reads_the_minibatch_in_place_from_the_buffer():
    """fabhip_flow_log_prob_tape_rows: batch row g = row rows[g] of the buffer - bit-identical to gathering first, on both tile
    shapes (through the one-op step's building blocks: the C ABI driven with ctypes)."""
    import ctypes as C
    from fab_torch_amd import _lib, _ops
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    lib.fabhip_flow_log_prob_tape_rows.argtypes = [C.POINTER(_lib.Flow), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                   C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fabhip_flow_tape_bytes.restype = C.c_size_t
    lib.fabhip_flow_tape_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    for (D, K, nodes) in ((32, 3, 10), (6, 3, 8)):
        flow = _flow(D, K, nodes, dev)
        packed, _, _, W = flow.native(need_inverse=False)
        bufx = torch.randn(5000, D, device=dev)
        rows = torch.randperm(5000, device=dev)[:300].contiguous()
        nb = lib.fabhip_flow_tape_bytes(D, K, W, 300)
        f = _lib.Flow(D, K, W, _lib.ptr(packed))
        for mode in (0, 16):
            with _ops.option(_ops.OPT_TAPE_TILES, mode):
                lq_a, tape_a = flow.log_prob_with_tape(bufx[rows])
                lq_b = torch.empty(300, device=dev)
                tape_b = torch.empty(nb // 4 + 64, device=dev)
                _lib.check(lib.fabhip_flow_log_prob_tape_rows(C.byref(f), _lib.ptr(bufx), _lib.ptr(rows), _lib.ptr(lq_b), None, 300,
                                                              _lib.ptr(tape_b), nb, _lib.stream_ptr()), "tape_rows")
                torch.cuda.synchronize()
                assert torch.equal(lq_a, lq_b)


def test_buffer_add_and_sample_ops_match_the_tensor_expressions():
    """fabhip::buffer_add = the ring write of prioritised_replay_buffer.py:71-85 (incl. the wrap); fabhip_buffer_sample with
    given uniforms = top-k of logit + Gumbel(u) as a set, ordered by the second set of uniforms."""
    import ctypes as C
    from fab_torch_amd import _lib, _ops
    dev = torch.device("cuda", 0)
    ops = _ops.load()
    L, D = 1000, 7
    bx, blw, blq = torch.zeros(L, D, device=dev), torch.zeros(L, device=dev), torch.zeros(L, device=dev)
    rx, rlw, rlq = bx.clone(), blw.clone(), blq.clone()
    start = 0
    for n in (300, 300, 300, 300, 17):
        x, lw, lq = torch.randn(n, D, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
        ops.buffer_add(x, lw, lq, start, bx, blw, blq)
        idx = (torch.arange(n, device=dev) + start) % L
        rx[idx], rlw[idx], rlq[idx] = x, lw, lq
        start = (start + n) % L
        assert torch.equal(bx, rx) and torch.equal(blw, rlw) and torch.equal(blq, rlq)
    lib = _lib.load()
    lib.fabhip_buffer_sample_workspace_bytes.restype = C.c_size_t
    lib.fabhip_buffer_sample_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.fabhip_buffer_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]
    for n, k in ((100000, 4096), (512000, 16384), (300, 300), (50, 1), (200000, 50000)):
        g = torch.Generator(device=dev).manual_seed(n)
        logw = torch.randn(n, device=dev, generator=g) * 3
        logw[::7] = -float("inf")
        u = torch.rand(n, device=dev, generator=g)
        r = torch.rand(4, device=dev, generator=g)
        nb = lib.fabhip_buffer_sample_workspace_bytes(n, k)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
        wsp = (ws.data_ptr() + 255) & ~255
        out = torch.empty(k, dtype=torch.int64, device=dev)
        _lib.check(lib.fabhip_buffer_sample(_lib.ptr(logw), _lib.ptr(u), _lib.ptr(r), n, k, _lib.ptr(out), C.c_void_p(wsp), nb,
                                            _lib.stream_ptr()), "buffer_sample")
        keys = -torch.log(-torch.log(u.clamp(min=torch.finfo(torch.float32).tiny))) + logw
        if k < n:
            kth = torch.topk(keys, k).values.min()
            assert bool((keys[out] >= kth).all())
        assert out.unique().numel() == k
        # the order: a bijection of the selection (keyed by r), another key = another order, the same key = the same order
        sel_sorted = torch.sort(out).values
        if k < n:
            assert torch.equal(sel_sorted, torch.sort(torch.topk(keys, k).indices).values) or torch.unique(keys).numel() < n
        out2, out3 = torch.empty_like(out), torch.empty_like(out)
        for o, rr in ((out2, r), (out3, torch.rand(4, device=dev, generator=g))):
            _lib.check(lib.fabhip_buffer_sample(_lib.ptr(logw), _lib.ptr(u), _lib.ptr(rr), n, k, _lib.ptr(o), C.c_void_p(wsp), nb,
                                                _lib.stream_ptr()), "buffer_sample")
        assert torch.equal(out, out2) and torch.equal(torch.sort(out3).values, sel_sorted)
        if k >= 300:
            assert not torch.equal(out, out3) and not torch.equal(out, sel_sorted)
            # no trace of the index order: the rank correlation between position and row index is that of a shuffle
            pos = torch.arange(k, device=dev, dtype=torch.float64)
            rk = torch.argsort(torch.argsort(out)).double()
            corr = float(((pos - pos.mean()) * (rk - rk.mean())).sum() / (pos.var(unbiased=False) * k))
            assert abs(corr) < 5.0 / np.sqrt(k)
    # the op (draws from torch's device generator): a set of k distinct valid rows, reproducible under a seed
    torch.manual_seed(5)
    a = ops.buffer_sample_indices(logw, 1)
    logw = torch.randn(70000, device=dev)
    torch.manual_seed(5); a = ops.buffer_sample_indices(logw, 2048)
    torch.manual_seed(5); b = ops.buffer_sample_indices(logw, 2048)
    assert torch.equal(a, b) and a.unique().numel() == 2048 and int(a.min()) >= 0 and int(a.max()) < 70000


@pytest.mark.parametrize("act_norm,clip", [(False, None), (False, 10.0), (True, None)])
def test_one_op_minibatch_step_equals_the_step_by_step_trainer(act_norm, clip):
    """PrioritisedBufferTrainer with every minibatch as ONE `fabhip::buffer_train_step` call against the same trainer stepping
    through the separate ops (tape, torch expressions for the loss weights, parameter gradients, FlatAdam, buffer.adjust): same
    draws, 4 iterations x 3 minibatches - loss, gradient norm, every parameter and the buffer's weights agree to 1e-5."""
    import fab_torch_amd as fa
    from fab_torch_amd.buffer import PrioritisedReplayBuffer
    dev = torch.device("cuda", 0)
    D, K, nodes, M, B = 6, 3, 40, 2, 256
    results = []
    for one_op in (False, True):
        torch.manual_seed(1)
        flow = fa.make_wrapped_normflow_realnvp(D, K, nodes, act_norm=act_norm).to(dev)
        target = fa.ManyWellEnergy(D)
        hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=3).to(dev)
        model = fa.FABModel(flow, target, M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
        ais = model.annealed_importance_sampler
        opt = fa.FlatAdam(flow, lr=1e-3)

        def init_sampler():
            pt, lw = ais.sample_and_log_weights(B, logging=False)
            return pt.x, lw, pt.log_q
        buf = PrioritisedReplayBuffer(D, 4096, 1024, init_sampler, device=dev)
        tr = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=2.0, n_batches_buffer_sampling=3, max_gradient_norm=5.0,
                                         w_adjust_max_clip=clip)
        tr.one_op_minibatch = one_op
        torch.manual_seed(7)
        infos = [tr.step(i + 1, B) for i in range(4)]
        results.append((infos, torch.cat([p.detach().reshape(-1) for p in flow.parameters()]).clone(), buf.buffer.log_w.clone(),
                        buf.buffer.log_q_old.clone(), tr.last_indices.clone(), int(opt.steps.item())))
    (ia, pa, lwa, lqa, idxa, sa), (ib, pb, lwb, lqb, idxb, sb) = results
    assert torch.equal(idxa, idxb) and sa == sb == 12
    for a, b in zip(ia, ib):
        for key in ("loss", "grad_norm", "w_adjust_mean", "w_adjust_min", "w_adjust_max", "log_q_x_mean", "sampled_log_w_mean",
                    "sampled_log_w_std"):
            assert abs(a[key] - b[key]) <= 1e-5 * max(1.0, abs(a[key])), (key, a[key], b[key])
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max())
    fin = torch.isfinite(lwa)
    assert torch.equal(fin, torch.isfinite(lwb))
    assert float((lwa[fin] - lwb[fin]).abs().max()) <= 1e-4 and float((lqa - lqb).abs().max()) <= 1e-4


def test_one_op_minibatch_step_skips_the_update_on_a_non_finite_loss_and_kills_the_rows():
    """A buffer row whose stored log q is far above the current one gives exp(+large) = inf weight: the loss is not finite, the
    optimiser must not move (reference :172-181) and the step counter must not advance; a row with a non-finite adjustment gets
    log_w = -inf (prioritised_replay_buffer.py:128-131), a finite one - however large - is added."""
    import fab_torch_amd as fa
    from fab_torch_amd import _ops
    dev = torch.device("cuda", 0)
    ops = _ops.load()
    D, K, nodes, B = 6, 3, 40, 64
    flow = _flow(D, K, nodes, dev)
    opt = fa.FlatAdam(flow, lr=1e-2)
    N = 500
    bx = torch.randn(N, D, device=dev)
    blw = torch.zeros(N, device=dev)
    blq = flow.log_prob(bx).detach().clone()
    blq[3] = float("nan")                                   # -> non-finite adjustment: the row is killed
    blq[5] += 500.0                                         # -> w = exp(500) = inf: the loss is not finite
    rows = torch.arange(B, device=dev)
    before = opt.theta.detach().clone()
    packed, _, _, W = flow.native(need_inverse=False)
    with torch.no_grad():
        lq, adj, stats = ops.buffer_train_step(flow._own_handle(), packed, D, K, W, False, bx, rows, blq, True, 2.0, 0.0, blw, blq,
                                               opt.theta.detach(), opt.m, opt.v, 1e-2, 0.9, 0.999, 1e-8, opt.steps, 5.0)
    st = stats.tolist()
    assert not np.isfinite(st[0]) and not np.isfinite(st[5])
    assert torch.equal(opt.theta.detach(), before) and int(opt.steps.item()) == 0
    # (row 5's adjustment, +500, is finite: the reference adds it, :124-127; only the NaN row is killed)
    assert float(blw[3]) == -float("inf") and abs(float(blw[5]) - 500.0) < 1e-2 and float(blw[4]) != -float("inf")
    assert bool(torch.isfinite(blw[6:B]).all()) and torch.equal(blw[B:], torch.zeros(N - B, device=dev))
