"""BASELINE cfg 3 with the flow family it names: ManyWell-32, spline flow 12 layers (hidden 256, 8 bins), 2048 chains,
12 intermediate distributions, HMC(5 leapfrogs) - through the generic plug-in path (spline kernels + elementwise
transition kernels).  Prints AIS samples/s and the time of one log_prob_and_grad / sample call."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa

DEV = "cuda"
D, L, H, M, B = 32, 12, 256, 12, 2048
LF, CIRC, TB = 5, (), 5.0
if os.environ.get("CFG") == "5":                       # BASELINE cfg 5's shape: alanine-dipeptide flow on a 60-D stand-in target
    import math
    D, M, B, LF = 60, 20, 4096, 10
    CIRC = (3, 7, 8, 12, 20, 21, 22, 30, 41, 45, 52, 59)
    TB = torch.full((D,), 5.0); TB[list(CIRC)] = math.pi
B = int(os.environ.get("B", B))
torch.manual_seed(0)
flow = fa.make_wrapped_normflow_spline(D, L, H, CIRC, TB).to(DEV).requires_grad_(False)
with torch.no_grad():
    for p in flow.parameters():
        if p.dim() == 2 and p.shape[0] % 25 == 0:
            p.add_(0.02 * torch.randn_like(p))
target = fa.ManyWellEnergy(D)
hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=LF).to(DEV)
ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)


def timeit(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


x, _ = flow.sample_and_log_prob((B,))
out = {"config": f"ManyWell-{D}, spline {L}x(hidden {H}, 8 bins, {len(CIRC)} circular), {B} chains, M={M}, HMC L={LF}"}
out["log_prob_and_grad_ms"] = 1e3 * timeit(lambda: flow.log_prob_and_grad(x), 50)
out["log_prob_ms"] = 1e3 * timeit(lambda: flow.log_prob(x), 50)
out["sample_ms"] = 1e3 * timeit(lambda: flow.sample_and_log_prob((B,)), 50)
t = timeit(lambda: ais.sample_and_log_weights(B), int(os.environ.get("N", 5)), warm=2)
out["ais_call_ms"] = 1e3 * t
out["ais_samples_per_s"] = B / t
out["n_flow_grad_evals_per_call"] = M * LF + 1            # one per leapfrog + the chain initialisation
with fa.fast_mode():                                   # bf16 conditioner GEMMs where they are the faster kernel (> 8 chains per CU)
    out["fast_log_prob_and_grad_ms"] = 1e3 * timeit(lambda: flow.log_prob_and_grad(x), 50)
    t = timeit(lambda: ais.sample_and_log_weights(B), int(os.environ.get("N", 5)), warm=2)
    out["fast_ais_call_ms"] = 1e3 * t
    out["fast_ais_samples_per_s"] = B / t
print(json.dumps(out))
