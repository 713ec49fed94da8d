"""Robustness sweep at sizes beyond the BASELINE configs: large batches, ragged batches, D = 64, both flow families."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa

DEV = "cuda"
rows = []
for name, D, K, nodes, M, B in (("realnvp mw32 65537 chains", 32, 10, 10, 4, 65537), ("realnvp D=64 W=512 4099 chains", 64, 4, 8, 3, 4099),
                                ("realnvp mw6 1 chain", 6, 3, 8, 2, 1), ("realnvp mw32 262144 chains", 32, 10, 10, 2, 262144)):
    torch.manual_seed(0)
    flow = fa.RealNVP(D, K, nodes).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=3).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pt, lw = ais.sample_and_log_weights(B)
    torch.cuda.synchronize()
    rows.append({"case": name, "ms": 1e3 * (time.perf_counter() - t0), "rows": int(pt.x.shape[0]), "finite": bool(torch.isfinite(lw).all()),
                 "ess": ais.get_logging_info()["ess_ais"], "mem_GB": torch.cuda.max_memory_allocated() / 1e9})
    print(json.dumps(rows[-1]), flush=True)
    del flow, ais, pt, lw
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
for name, D, L, H, circ, M, B in (("spline 32-D 16385 chains", 32, 12, 256, (), 3, 16385), ("spline 60-D 12 circ 65536 chains", 60, 12, 256, (3, 7, 8, 12, 20, 21, 22, 30, 41, 45, 52, 59), 2, 65536),
                                  ("spline 4-D hidden 16 3 chains", 4, 2, 16, (1,), 2, 3)):
    import math
    torch.manual_seed(0)
    tb = torch.full((D,), 5.0); tb[list(circ)] = math.pi
    flow = fa.make_wrapped_normflow_spline(D, L, H, circ, tb).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=3).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pt, lw = ais.sample_and_log_weights(B)
    x, lq = flow.sample_and_log_prob((B,))
    rt = float((flow.log_prob(x) - lq).abs().max())
    torch.cuda.synchronize()
    rows.append({"case": name, "ms": 1e3 * (time.perf_counter() - t0), "rows": int(pt.x.shape[0]), "finite": bool(torch.isfinite(lw).all()),
                 "ess": ais.get_logging_info()["ess_ais"], "sample_logprob_roundtrip": rt, "mem_GB": torch.cuda.max_memory_allocated() / 1e9})
    print(json.dumps(rows[-1]), flush=True)
    del flow, ais, pt, lw
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
