"""Profiling driver: N density + gradient evaluations of the cfg-3 spline flow (k_spline_logprob_r8 / k_spline_logprob).
Usage (GPU box):  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_spline_logprob -- python tools/prof_spline.py [n] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
torch.manual_seed(0)
flow = fa.make_wrapped_normflow_spline(32, 12, 256, (), 5.0).to("cuda").requires_grad_(False)
x = torch.randn(B, 32, device="cuda")
for _ in range(n):
    flow.log_prob_and_grad(x)
torch.cuda.synchronize()
print("done", n, B)
