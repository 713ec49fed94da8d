"""End-to-end FAB training demo on the GPU (prioritised buffer, alpha = 2) — ManyWell target, HIP AIS + HIP flow
training path + FlatAdam.  Records the evaluation metrics of `FABModel.get_eval_info` along the way.
Usage (GPU box): python tools/train_demo.py [--dim 6] [--iters 1500] [--batch 512] > curve.json"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=6)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--M", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--evals", type=int, default=10)
    ap.add_argument("--flow", choices=("realnvp", "spline"), default="realnvp")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--fast", action="store_true", help="AIS transitions in fast mode (bf16 W x W GEMMs) during training; "
                                                          "evaluation always runs the fp32 parity path")
    args = ap.parse_args()
    torch.manual_seed(0)
    D = args.dim
    if args.flow == "spline":       # the alanine-dipeptide flow family on the ManyWell target (BASELINE cfg 3), torch Adam
        flow = fa.make_wrapped_normflow_spline(D, args.layers, args.hidden, (), 5.0).to(DEV)
    else:
        flow = fa.make_wrapped_normflow_realnvp(D, n_flow_layers=args.layers, layer_nodes_per_dim=40 if D <= 8 else 10,
                                                act_norm=False).to(DEV)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(args.M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=1.0,
                                   n_outer=1, L=5).to(DEV)
    model = fa.FABModel(flow, target, args.M, alpha=2.0, transition_operator=hmc, loss_type="fab_alpha_div")
    ais = model.annealed_importance_sampler
    opt = fa.FlatAdam(flow, lr=args.lr)          # RealNVP: tape + flat-image kernels; spline: autograd + fused clip / Adam

    def init_sampler():
        pt, lw = ais.sample_and_log_weights(args.batch, logging=False)
        return pt.x, lw, pt.log_q

    buf = fa.PrioritisedReplayBuffer(D, 200 * args.batch, 20 * args.batch, init_sampler, device=DEV)
    trainer = fa.PrioritisedBufferTrainer(model, opt, buf, alpha=2.0, n_batches_buffer_sampling=4,
                                          max_gradient_norm=100.0, w_adjust_max_clip=10.0)
    curve = []
    chunk = max(args.iters // args.evals, 1)
    done = 0
    t_train = 0.0
    while done < args.iters:
        n = min(chunk, args.iters - done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with fa.fast_mode(args.fast):
            hist = trainer.run(done + n, args.batch, start_iter=done)
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        done += n
        ev = model.get_eval_info(outer_batch_size=8192, inner_batch_size=2048)
        ev.update(iteration=done, train_seconds=t_train, ess_ais_train=hist[-1]["ess_ais"], loss=hist[-1]["loss"])
        curve.append(ev)
        print(json.dumps(ev), file=sys.stderr)
    print(json.dumps({"config": vars(args), "ms_per_iteration": t_train / args.iters * 1e3, "curve": curve}, indent=1))


if __name__ == "__main__":
    main()
