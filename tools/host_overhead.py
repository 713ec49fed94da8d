"""Host-side cost of one headline AIS call, piece by piece (the GPU is idle while the host prepares the next call: the call
returns a data-dependent number of rows, so it ends in a device->host read).  perf_counter around each piece, the device
synchronised before each call so that every figure is pure host time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                                        # noqa: E402
import fab_torch_amd as fa                                                                          # noqa: E402
from fab_torch_amd import _ops                                                                      # noqa: E402

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
ops = _ops.load()
acc = {}


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


N = 200
f32 = dict(dtype=torch.float32, device=dev)
for _ in range(N):
    torch.cuda.synchronize()
    t = time.perf_counter()
    eps0 = torch.randn((B, D), **f32)
    noise_a = torch.randn((M, 1, B, D), **f32)
    noise_b = torch.empty((M, 1, B), **f32).exponential_(1.0)
    t = tick("3 noise draws", t)
    fl, tg = ais._native_parts()
    t = tick("_native_parts", t)
    fargs = fl.native()
    t = tick("flow.native()", t)
    targs = tg.native_target()
    t = tick("target.native_target()", t)
    betas = [float(b) for b in ais.B_space]
    t = tick("betas list", t)
    op = ais.transition_operator
    out = ops.ais_run(*fargs, *targs, betas, float(ais.alpha), False, _ops.TRANSITION_HMC, eps0, noise_a, noise_b, op.epsilons,
                      op.common_epsilon, op.mass_vector, 1, op.L, float(op.max_grad), float(op.target_p_accept), True,
                      op._p_accept_first, op._p_accept_last, op._dist_first, op._dist_last, False, _ops.precision_of(fl))
    t = tick("ops.ais_run (host side)", t)
    torch.cuda.synchronize()
    t = time.perf_counter()
    host = torch.cat([out[6].float(), out[7][:6]]).cpu()
    t = tick("cat + .cpu() on an idle device", t)
    n_init, n_end = int(host[0]), int(host[1])
    st = host[2:]
    _ = (float(st[0]), float(st[3]), float(st[4]))
    t = tick("host parse", t)
for k, v in acc.items():
    print(f"{k:34s} {v / N * 1e6:8.1f} us")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    ais.sample_and_log_weights(B)
torch.cuda.synchronize()
print(f"sample_and_log_weights: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per call")
