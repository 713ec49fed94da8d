"""Fast mode (bf16 W x W GEMMs in the transition kernels) against the fp32 parity mode on the headline workload:
ManyWell-32, RealNVP 10 x (16-320-320-32), 1024 chains, M = 8, HMC L = 5.  Prints the density deviation, the AIS rate of
both modes and the ESS / log Z each mode reports on the same noise."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa
from bench import build_flow_state      # the bench's seeded headline flow

DEV = "cuda"
D, K, M, B = 32, 10, 8, int(os.environ.get("B", 1024))
flow = build_flow_state(0).to(DEV).requires_grad_(False)
target = fa.ManyWellEnergy(D)
hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=5,
                               eval_mode=True).to(DEV)
ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
torch.manual_seed(0)
eps0 = torch.randn(B, D, device=DEV); na = torch.randn(M, 1, B, D, device=DEV)
nb = torch.empty(M, 1, B, device=DEV).exponential_()


def run():
    return ais.sample_and_log_weights(B, eps0=eps0, noise_a=na, noise_b=nb)


def rate(n=30):
    for _ in range(5):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


out = {"config": f"ManyWell-{D}, RealNVP {K}x(16-320-320-32), {B} chains, M={M}, HMC L=5"}
x, _ = flow.sample_and_log_prob((B,))
pt32 = fa.create_point(x, flow, target, with_grad=True)
with fa.fast_mode():
    ptf = fa.create_point(x, flow, target, with_grad=True)
dq = (ptf.log_q - pt32.log_q).abs()
out["log_q_abs_dev"] = {"max": float(dq.max()), "mean": float(dq.mean()), "log_q_scale": float(pt32.log_q.abs().mean())}
gn = pt32.grad_log_q.norm(dim=1)
out["grad_rel_l2_dev"] = {"max": float(((ptf.grad_log_q - pt32.grad_log_q).norm(dim=1) / gn).max()),
                          "mean": float(((ptf.grad_log_q - pt32.grad_log_q).norm(dim=1) / gn).mean())}
p32, lw32 = run()
i32 = ais.get_logging_info()
out["fp32"] = {"samples_per_s": rate(), "ess_ais": i32["ess_ais"], "log_Z": i32.get("log_Z_ais", None)}
with fa.fast_mode():
    pf, lwf = run()
    i_f = ais.get_logging_info()
    out["fast"] = {"samples_per_s": rate(), "ess_ais": i_f["ess_ais"], "log_Z": i_f.get("log_Z_ais", None)}
same = ((pf.x - p32.x).abs().max(1).values < 1e-2).float().mean()
out["chains_on_the_fp32_trajectory"] = float(same)
out["speedup"] = out["fast"]["samples_per_s"] / out["fp32"]["samples_per_s"]
print(json.dumps(out))
