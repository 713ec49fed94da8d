"""Profiling driver: N HMC transitions (the dominant kernel k_hmc_step) on the headline config.
Usage (GPU box):  rocprofv3 --kernel-trace --stats -d out -- python tools/prof_hmc.py [n] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fab_torch_amd as fa  # noqa: E402
from fab_torch_amd.transition_operators import create_point  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.B_PER_GPU
dev = torch.device("cuda", 0)
flow = bench.build_flow_state(0).to(dev).requires_grad_(False)
target = fa.ManyWellEnergy(bench.D)
hmc = fa.HamiltonianMonteCarlo(bench.M, bench.D, flow.log_prob, target.log_prob, alpha=bench.ALPHA, p_target=False,
                               epsilon=bench.EPS_INIT, n_outer=1, L=bench.L, eval_mode=True).to(dev)
if os.environ.get("FAST") == "1":                       # profile k_hmc_step_fast (bf16 W x W GEMMs) instead
    fa.fast_mode(True)
x0, _ = flow.native_sample(torch.randn(B, bench.D, device=dev))
pt = create_point(x0, flow, target, with_grad=True)
for _ in range(n):
    hmc.transition(pt, 4, 0.444)
torch.cuda.synchronize()
print("done", n, B)
