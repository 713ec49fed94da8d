"""Duration of the tail of a chain phase (compaction + log_p - log_q + ESS / log Z) as ONE launch (k_tail_small) against the
separate kernels, per batch size: HIP events around whole AIS calls of a 1-layer flow with one 1-leapfrog transition (the tails
are a visible share of such a call), FABHIP_OPT_FUSED_TAIL on / off.  Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa                                                                          # noqa: E402
from fab_torch_amd import _ops                                                                      # noqa: E402

dev = "cuda"
D = 32
ops = _ops.load()
flow = fa.RealNVP(D, 1, 4).to(dev).requires_grad_(False)
target = fa.ManyWellEnergy(D)
for B in (1024, 2048, 4096, 8192):
    row = []
    for mode in (1, 0):
        ops.set_option(_ops.OPT_FUSED_TAIL, mode)
        hmc = fa.HamiltonianMonteCarlo(1, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, n_outer=1,
                                       L=1).to(dev)
        ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=2.0, n_intermediate_distributions=1)
        eps0 = torch.randn(B, D, device=dev); na = torch.randn(1, 1, B, D, device=dev); nb = torch.empty(1, 1, B, device=dev).exponential_()
        for _ in range(20):
            ais.run(B, eps0, na, nb)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
        ev[0].record()
        for i in range(100):
            ais.run(B, eps0, na, nb)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(100))
        row.append(ms[50] * 1e3)
    ops.set_option(_ops.OPT_FUSED_TAIL, 1)
    print(f"B={B}: AIS call with one-launch tails {row[0]:.1f} us, with separate kernels {row[1]:.1f} us")
