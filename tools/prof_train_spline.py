"""Phase times of one FAB training iteration with the spline flow (ManyWell-32, 12 layers x hidden 256, batch 1024, M=4)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa

DEV = "cuda"
D, L, H, M, B = 32, 12, 256, 4, 1024
torch.manual_seed(0)
flow = fa.make_wrapped_normflow_spline(D, L, H, (), 5.0).to(DEV)
target = fa.ManyWellEnergy(D)
hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=5).to(DEV)
ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
opt = torch.optim.Adam(flow.parameters(), lr=1e-4)


def t(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


x, _ = flow.sample_and_log_prob((B,))
x = x.detach()
out = {"ais_call_ms": t(lambda: ais.sample_and_log_weights(B))}
out["log_prob_fwd_tape_ms"] = t(lambda: flow.log_prob(x))


def fb():
    opt.zero_grad()
    (-flow.log_prob(x).mean()).backward()


out["fwd_bwd_ms"] = t(fb)


def full():
    fb(); torch.nn.utils.clip_grad_norm_(flow.parameters(), 100.0); opt.step()


out["fwd_bwd_clip_adam_ms"] = t(full)
fopt = fa.FlatAdam(flow, lr=1e-4)


def full_flat():
    fopt.zero_grad()
    (-flow.log_prob(x).mean()).backward()
    fopt.step(max_grad_norm=100.0)


out["fwd_bwd_flat_adam_ms"] = t(full_flat)
p0 = next(flow.parameters())
with torch.no_grad():
    out["pack_ms"] = t(lambda: (p0.add_(0.0), flow.native()))                  # re-pack after an optimiser step
print(json.dumps(out))
