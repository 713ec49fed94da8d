"""Phase breakdown of one FAB training iteration with the prioritised buffer on the GPU (SURVEY.md section 8f
"next" rows): AIS call, buffer.add, Gumbel-top-k sampling, the 8 minibatch (log_prob fwd, backward, clip, Adam,
buffer.adjust) steps.  ManyWell-32 training shape of the reference config (batch 2048, M=4, L=5, buffer 512000,
8 minibatches).  Prints one JSON object.   Usage (GPU box): python tools/prof_train.py [--iters 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa  # noqa: E402
from fab_torch_amd.buffer import PrioritisedReplayBuffer  # noqa: E402

DEV = torch.device("cuda", 0)


class Phases:
    def __init__(self):
        self.t = {}

    def time(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--buffer", type=int, default=512000)
    ap.add_argument("--min-buffer", type=int, default=65536)
    ap.add_argument("--path", default="hip", choices=["hip", "torch"],
                    help="flow.log_prob training path: HIP tape + parameter-gradient kernels, or torch autograd")
    ap.add_argument("--optim", default="flat", choices=["flat", "torch"],
                    help="FlatAdam (fused clip + Adam kernels) or clip_grad_norm_ + torch.optim.Adam")
    args = ap.parse_args()
    D, M, L, NB, alpha = 32, 4, 5, 8, 2.0
    torch.manual_seed(0)
    flow = fa.make_wrapped_normflow_realnvp(D, n_flow_layers=10, layer_nodes_per_dim=10, act_norm=False).to(DEV)
    if args.path == "torch":          # comparison arm only: the ATen expression of tests/aten_reference.py
        import sys, os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import aten_reference
        hip_log_prob = flow.log_prob
        flow.log_prob = lambda x: (aten_reference.log_prob(flow, x) if torch.is_grad_enabled() else hip_log_prob(x))
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=alpha, p_target=False,
                                   epsilon=0.2, n_outer=1, L=L).to(DEV)
    model = fa.FABModel(flow, target, M, alpha=alpha, transition_operator=hmc, loss_type="fab_alpha_div")
    ais = model.annealed_importance_sampler
    opt = fa.FlatAdam(flow, lr=3e-4) if args.optim == "flat" else torch.optim.Adam(flow.parameters(), lr=3e-4)

    def init_sampler():
        pt, lw = ais.sample_and_log_weights(args.batch, logging=False)
        return pt.x, lw, pt.log_q

    t0 = time.perf_counter()
    buf = PrioritisedReplayBuffer(D, args.buffer, args.min_buffer, init_sampler, device=DEV)
    torch.cuda.synchronize()
    fill_s = time.perf_counter() - t0

    ph = Phases()
    for it in range(args.iters + 2):
        if it == 2:
            ph.t.clear()                                  # two warm-up iterations
        pt, lw = ph.time("ais", lambda: ais.sample_and_log_weights(args.batch))
        ph.time("buffer_add", lambda: buf.add(pt.x.detach(), lw.detach(), pt.log_q.detach()))
        mini = ph.time("buffer_sample", lambda: buf.sample_n_batches(args.batch, NB))
        for (x, log_w, log_q_old, idx) in mini:
            opt.zero_grad()
            log_q = ph.time("flow_log_prob_fwd", lambda: flow.log_prob(x))

            def loss_bwd():
                adj = (1 - alpha) * (log_q.detach() - log_q_old)
                loss = -torch.mean(torch.exp(adj) * log_q)
                loss.backward()
                return adj
            adj = ph.time("loss_backward", loss_bwd)
            if args.optim == "flat":
                ph.time("clip_adam", lambda: opt.step(max_grad_norm=100.0))
            else:
                ph.time("clip_adam", lambda: (torch.nn.utils.clip_grad_norm_(flow.parameters(), 100.0), opt.step()))
            ph.time("buffer_adjust", lambda: buf.adjust(adj, log_q.detach(), idx))
    n = args.iters
    per_iter = {k: v / n * 1e3 for k, v in ph.t.items()}
    per_iter["total"] = sum(per_iter.values())
    # the same iteration through the trainer, no per-phase synchronisation (one .item() sync per iteration)
    trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=alpha, n_batches_buffer_sampling=NB,
                                          max_gradient_norm=100.0, w_adjust_max_clip=10.0)
    trainer.run(2, args.batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.run(n, args.batch)
    torch.cuda.synchronize()
    per_iter["trainer_end_to_end"] = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"flow_train_path": args.path, "optimiser": args.optim,
                      "config": f"ManyWell-32 training iteration: batch {args.batch}, M={M}, L={L}, buffer "
                                f"{args.buffer}, {NB} minibatches", "buffer_fill_s": fill_s,
                      "ms_per_iteration": per_iter}))


if __name__ == "__main__":
    main()
