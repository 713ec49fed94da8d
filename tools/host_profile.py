"""cProfile of the host side of the headline AIS call (which Python lines the GPU waits for between two calls)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                                        # noqa: E402
import fab_torch_amd as fa                                                                          # noqa: E402

dev = torch.device("cuda:0")
B, D, M, L = bench.B_PER_GPU, bench.D, bench.M, bench.L
flow = bench.build_flow_state(0).to(dev).requires_grad_(False)
target = fa.ManyWellEnergy(D)
hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=bench.ALPHA, p_target=False,
                               epsilon=bench.EPS_INIT, n_outer=1, L=L).to(dev)
ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=bench.ALPHA,
                                   n_intermediate_distributions=M)
for _ in range(100):
    ais.sample_and_log_weights(B)
torch.cuda.synchronize()
N = 300
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    ais.sample_and_log_weights(B)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / N * 1e6, ct / N * 1e6, nc / N, f"{os.path.basename(fn)}:{line}:{name}"))
rows.sort(reverse=True)
print(f"{'tottime us':>10} {'cumtime us':>10} {'calls':>6}  function (per AIS call)")
for tt, ct, nc, nm in rows[:32]:
    print(f"{tt:10.1f} {ct:10.1f} {nc:6.1f}  {nm}")
