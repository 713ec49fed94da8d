"""One FAB training iteration with the prioritised buffer on the reference's ManyWell-32 recipe, timed end to end and by phase.

Recipe (`/root/reference/experiments/config/many_well.yaml`: flow 10 x (16-320-320-32 + InvertibleAffine), batch 2048, M = 4,
HMC L = 5, alpha = 2, 8 minibatches per iteration (`n_batches_buffer_sampling`), buffer 512 000 / min 65 536, lr 3e-4,
max_grad_norm 100, no weight clipping) through `fab_torch_amd.PrioritisedBufferTrainer.step` =
`fab/train_with_prioritised_buffer.py:138-216`.  Timing: `iteration_ms` = N back-to-back `trainer.step` calls between two
device synchronisations (every step ends with the reference's own `.item()` reads, so nothing is hidden behind the host);
`ais_ms` / `train_ms` = HIP events around the AIS call and around everything after it (buffer add, sampling, 8 minibatch steps)
inside the same steps.  `measure()` is what bench.py's `trainer_iteration` row calls; run stand-alone it prints one JSON object
(and under `rocprofv3 --kernel-trace --stats` gives the per-kernel rows of profiles/r6/)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

D, K_LAYERS, NODES, BATCH, M, L, NB, ALPHA = 32, 10, 10, 2048, 4, 5, 8, 2.0
BUFFER, MIN_BUFFER, LR, MAX_GRAD_NORM = 512000, 65536, 3e-4, 100.0
F_FWD = K_LAYERS * 2 * (16 * 320 + 320 * 320 + 2 * 320 * 16) + 2 * K_LAYERS * D * D      # flop per sample and sweep
# parameter gradients: per layer dW1 (320 x 16), dW2 (320 x 320), dW3 (32 x 320), dW' (32 x 32), 2 flop per sample and entry
F_PGRAD = K_LAYERS * 2 * (320 * 16 + 320 * 320 + 32 * 320 + 32 * 32)
PEAK_FP32_MFMA_TFLOPS = 157.3


def build(dev, seed=0, eps_init=0.2):
    import fab_torch_amd as fa
    from fab_torch_amd.buffer import PrioritisedReplayBuffer
    torch.manual_seed(seed)
    flow = fa.make_wrapped_normflow_realnvp(D, n_flow_layers=K_LAYERS, layer_nodes_per_dim=NODES, act_norm=False).to(dev)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=ALPHA, p_target=False,
                                   epsilon=eps_init, n_outer=1, L=L).to(dev)
    model = fa.FABModel(flow, target, M, alpha=ALPHA, transition_operator=hmc, loss_type="fab_alpha_div")
    ais = model.annealed_importance_sampler
    opt = fa.FlatAdam(flow, lr=LR)

    def init_sampler():
        pt, lw = ais.sample_and_log_weights(BATCH, logging=False)
        return pt.x, lw, pt.log_q

    buf = PrioritisedReplayBuffer(D, BUFFER, MIN_BUFFER, init_sampler, device=dev)
    trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=ALPHA, n_batches_buffer_sampling=NB,
                                          max_gradient_norm=MAX_GRAD_NORM, w_adjust_max_clip=None)
    return trainer


def measure(dev, iters=8, warm=3):
    """(Few iterations on purpose: from a fresh flow this recipe in fp32 leaves the finite range around iteration 14 - an
    |x| ~ 40 AIS sample where the flow's density underflows enters the buffer - in the reference's own arithmetic as well
    (oracle/train.py on the CPU: same losses up to there, NaN at iteration 13); the skip branches of
    train_with_prioritised_buffer.py:169-179 then take over.  The timed iterations are the healthy ones before that; the
    launches of a skipped step are the same.)"""
    trainer = build(dev)
    ais = trainer.model.annealed_importance_sampler
    for i in range(warm):
        trainer.step(i + 1, BATCH)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(iters):
        info = trainer.step(warm + i + 1, BATCH)
    torch.cuda.synchronize(dev)
    it_ms = (time.perf_counter() - t0) / iters * 1e3
    # the AIS call of the iteration alone (same sampler, same state), HIP events on the ops' stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * iters)]
    for i in range(iters):
        ev[2 * i].record()
        ais.sample_and_log_weights(BATCH)
        ev[2 * i + 1].record()
    torch.cuda.synchronize(dev)
    ais_ms = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(iters))[iters // 2]
    out = {"workload": "trainer_iteration: ManyWell-32, RealNVP 10x(16-320-320-32)+InvAffine, batch 2048, M = 4, HMC L = 5, "
                       "8 minibatches of 2048 from a 512 000-entry prioritised buffer, alpha = 2, FlatAdam lr 3e-4, max_grad_norm 100 "
                       "(experiments/config/many_well.yaml through fab/train_with_prioritised_buffer.py:138-216)",
           "iterations_timed": iters, "iterations_warmup": warm, "iteration_ms": it_ms, "ais_ms": ais_ms, "train_ms": it_ms - ais_ms, "iterations_per_s": 1e3 / it_ms,
           "minibatch_us": (it_ms - ais_ms) / NB * 1e3, "loss": info["loss"], "grad_norm": info["grad_norm"],
           "ess_ais": info.get("ess_ais"),
           "flop_per_minibatch": {"tape_forward_reverse": 2 * BATCH * F_FWD, "param_grad": BATCH * F_PGRAD}}
    return out, trainer


def _graph_time(fn, dev, n=20, reps=10):
    """Seconds per call of `fn` as the GPU executes it: `reps` calls are captured into ONE HIP graph (their launches back to back,
    no host work) and the graph is replayed n times between two HIP events - a Python-level loop of op calls measures the host
    (one op call costs more host time than these kernels take), a one-call graph its own launch latency."""
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e-3 / (n * reps)


def kernel_rows(trainer, dev, n=20):
    """GPU time of the two training computations on a 2048-row minibatch (HIP-graph replays, see _graph_time) and of one whole
    minibatch step (fabhip::buffer_train_step: pack, tape, weights, gradients, Adam)."""
    from fab_torch_amd import _ops
    flow, opt, buf = trainer.model.flow, trainer.optimizer, trainer.buffer
    x = torch.randn(BATCH, D, device=dev)
    coef = torch.full((BATCH,), -1.0 / BATCH, device=dev)
    with torch.no_grad():
        # (one call per graph: the 118 MB tape is then the same allocation at every replay, as the trainer's own workspace is
        #  - four tapes in rotation fall out of the 256 MB memory-side cache and the kernel's stores take 35 us longer)
        t_tape = _graph_time(lambda: flow.log_prob_with_tape(x), dev, 20, 1)
        _, tape = flow.log_prob_with_tape(x)
        t_pg = _graph_time(lambda: flow.param_grad_flat(tape, coef), dev, 5)

        def pair():                                  # the two in the order the trainer runs them
            _, tp = flow.log_prob_with_tape(x)
            flow.param_grad_flat(tp, coef)
        t_pair = _graph_time(pair, dev, 5, 4)
        ops = _ops.load()
        packed, Dd, K, W = flow.native(need_inverse=False)
        rows = torch.randperm(buf.current_index if not buf.is_full else buf.max_length, device=dev)[:BATCH].contiguous()
        grp = opt.param_groups[0]
        theta = opt.theta.detach()
        blw, blq = buf.buffer.log_w.clone(), buf.buffer.log_q_old.clone()       # (the timing replays must not walk the real buffer)
        th, m, v, steps = theta.clone(), opt.m.clone(), opt.v.clone(), opt.steps.clone()

        def one_step():
            ops.buffer_train_step(flow._own_handle(), packed, Dd, K, W, True, buf.buffer.x, rows, blq, True, ALPHA, 0.0, blw, blq,
                                  theta, opt.m, opt.v, float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]),
                                  float(grp["eps"]), opt.steps, MAX_GRAD_NORM)
        t_step = _graph_time(one_step, dev, 5, 4)
        theta.copy_(th); opt.m.copy_(m); opt.v.copy_(v); opt.steps.copy_(steps)    # undo the timing replays' optimiser steps
        flow._packed_key = None
    out = {}
    for name, kern, t, flop in (("tape_forward_reverse", "k_flow_log_prob_tape_r8<5> (8 chains per workgroup, 256 workgroups)", t_tape,
                                 2 * BATCH * F_FWD),
                                ("param_grad", "k_pgrad_tiles<0> + k_pgrad_tiles<1> (one workgroup per output tile) + k_affine_grads",
                                 t_pg, BATCH * F_PGRAD)):
        ach = flop / t / 1e12
        out[name] = {"kernel": kern, "us_per_call": t * 1e6, "flop": flop, "achieved_TFLOPs": ach,
                     "frac_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS,
                     "timing": "HIP-graph replays of the op between two HIP events (GPU time of its launches, no host gaps); "
                               "per-kernel rows: profiles/r6/trainer_kernel_stats_rocprofv3.csv"}
    out["tape_then_param_grad"] = {"us_per_call": t_pair * 1e6, "flop": 2 * BATCH * F_FWD + BATCH * F_PGRAD,
                                   "frac_fp32_mfma_peak": (2 * BATCH * F_FWD + BATCH * F_PGRAD) / t_pair / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                   "note": "the two ops alternating as in the trainer (the tape kernel replayed back to back on its own "
                                           "runs ~30 us slower than inside an iteration: 165 us in profiles/r6/trainer_iteration_timeline.txt)"}
    out["minibatch_step"] = {"op": "fabhip::buffer_train_step (training pack, tape, loss weights + buffer.adjust, parameter "
                                   "gradients, LU chain rule, clipped Adam)", "us_per_call": t_step * 1e6,
                             "timing": "HIP-graph replays"}
    return out


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    iters = int(os.environ.get("ITERS", "8"))
    out, trainer = measure(dev, iters=iters)
    out["kernels"] = kernel_rows(trainer, dev)
    print(json.dumps(out))
