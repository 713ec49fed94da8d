"""What one slab all-gather costs the tuned sharded call when it is issued through RCCL on the compute stream (one-rank
communicator: all a 1-GPU box can run - RCCL refuses two ranks on one device, tools/experiments/rccl_two_ranks_one_gpu.py).
Headline shape (1024 chains, M = 8): fused single-device call vs `ais_sharded_tuned` over a one-rank "nccl" group (M real RCCL
all-gathers of the 129-float slab + M `fabhip_hmc_adapt_gathered` launches per call).  Prints ms per call and us per collective."""
import datetime
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                                        # noqa: E402
import fab_torch_amd as fa                                                                          # noqa: E402
from fab_torch_amd import parallel                                                                  # noqa: E402

dev = torch.device("cuda:0")
B, D, M, L = bench.B_PER_GPU, bench.D, bench.M, bench.L


def sampler():
    flow = bench.build_flow_state(0).to(dev).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=bench.ALPHA, p_target=False,
                                   epsilon=bench.EPS_INIT, n_outer=1, L=L).to(dev)
    return fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=bench.ALPHA,
                                        n_intermediate_distributions=M)


def timed(fn, n=60, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=dev)
ais_f, ais_s = sampler(), sampler()
be = parallel.HipShardBackend(ais_s)
t_fused = timed(lambda: ais_f.sample_and_log_weights(B))
n_coll = be.run_tuned(B, None)[2]
t_one = timed(lambda: be.run_tuned(B, None))
x = torch.ones(129, device=dev); out = torch.empty(129, device=dev)
t_ag = timed(lambda: dist.all_gather_into_tensor(out, x), n=400, warm=50)
print(f"fused single-device call {t_fused:.3f} ms | one-op tuned call over a one-rank RCCL group {t_one:.3f} ms, {n_coll} collectives "
      f"-> {(t_one - t_fused) / max(n_coll, 1) * 1e3:.1f} us per transition for (slab all-gather + rule launch, adaptation out of the "
      f"transition kernel) | a bare 129-float all_gather_into_tensor back to back: {t_ag * 1e3:.1f} us")
dist.destroy_process_group()
