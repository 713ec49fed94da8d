"""Median HIP-event time of one HMC transition (k_hmc_step + k_hmc_adapt) at the headline flow shape for a list
of chain counts.  Usage (GPU box): python tools/time_hmc.py 1024 4096 ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fab_torch_amd as fa  # noqa: E402
from fab_torch_amd.transition_operators import create_point  # noqa: E402

dev = torch.device("cuda", 0)
flow = bench.build_flow_state(0).to(dev).requires_grad_(False)
target = fa.ManyWellEnergy(bench.D)
hmc = fa.HamiltonianMonteCarlo(bench.M, bench.D, flow.log_prob, target.log_prob, alpha=bench.ALPHA, p_target=False,
                               epsilon=bench.EPS_INIT, n_outer=1, L=bench.L, eval_mode=True).to(dev)
for B in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    x0, _ = flow.native_sample(torch.randn(B, bench.D, device=dev))
    pt = create_point(x0, flow, target, with_grad=True)
    for _ in range(3):
        hmc.transition(pt, 4, 0.444)
    n = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        hmc.transition(pt, 4, 0.444)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]
    flop = B * bench.L * 2 * bench.F_FWD
    print(f"B={B}: {ms:.4f} ms/transition, {flop / ms / 1e9:.1f} TFLOP/s nominal")
