"""Secondary measurements (not the driver's bench line): resample-scan HBM roofline at large N, ESS
kernel, and AIS throughput on the other BASELINE configs.  Prints one JSON object.
Usage (GPU box): python tools/bench_extra.py [--quick]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa  # noqa: E402
from fab_torch_amd import _lib  # noqa: E402

DEV = torch.device("cuda", 0)
HBM_PEAK = 8.0e12


def ev_time(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ms[n // 2] * 1e-3


def resample_roofline(N, D=8):
    g = torch.Generator(device=DEV).manual_seed(0)
    lw = torch.randn(N, device=DEV, generator=g) * 3
    u = torch.rand(N, dtype=torch.float64, device=DEV, generator=g)
    x = torch.randn(N, D, device=DEV, generator=g)
    out = {"N": N, "D": D}
    t_sys = ev_time(lambda: fa.systematic_indices(lw, u0=0.3))
    t_mul = ev_time(lambda: fa.multinomial_indices(lw, u=u))
    idx = fa.systematic_indices(lw, u0=0.3)
    t_gat = ev_time(lambda: fa.gather_rows(x, idx))
    t_ess = ev_time(lambda: fa.ess_and_log_z(lw))
    # algorithmic bytes: log_w 4N (max pass) + 4N (scan read) + 8N (CDF write) ; search reads the
    # CDF (8N sorted / random) and writes idx 8N ; gather 2*4*N*D + 8N
    out["systematic_s"] = t_sys
    out["systematic_GBps"] = (4 * N + 4 * N + 8 * N + 8 * N + 8 * N) / t_sys / 1e9
    out["multinomial_s"] = t_mul
    out["multinomial_GBps"] = (4 * N + 4 * N + 8 * N + 8 * N + 8 * N + 8 * N) / t_mul / 1e9
    out["gather_s"] = t_gat
    out["gather_GBps"] = (8 * N + 2 * 4 * N * D) / t_gat / 1e9
    out["ess_s"] = t_ess
    out["ess_GBps"] = 4 * N / t_ess / 1e9
    for k in ("systematic", "multinomial", "gather", "ess"):
        out[k + "_frac_hbm"] = out[k + "_GBps"] * 1e9 / HBM_PEAK
    return out


def ais_config(name, D, K, nodes, B, M, kind, L=5, n_inner=1, eps=0.2, steps=10):
    torch.manual_seed(0)
    flow = fa.RealNVP(D, K, nodes)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.copy_(torch.randn(l3.weight.shape, generator=g) * 0.01)
    flow = flow.to(DEV).requires_grad_(False)
    if kind == "hmc":
        target = fa.ManyWellEnergy(D)
        op = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=eps,
                                      n_outer=n_inner, L=L).to(DEV)
    else:
        torch.manual_seed(0)
        target = fa.GMM(D, 40, 40.0, 1.0)
        op = fa.Metropolis(M, D, flow.log_prob, target.log_prob, n_updates=n_inner, alpha=2.0, p_target=False,
                           max_step_size=5.0, min_step_size=5.0, adjust_step_size=False).to(DEV)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, op, False, 2.0, M)
    for _ in range(3):
        ais.sample_and_log_weights(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    while done < steps or time.perf_counter() - t0 < 0.25:      # sub-millisecond configs need many calls
        ais.sample_and_log_weights(B)
        done += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / done
    info = ais.get_logging_info()
    return {"config": name, "B": B, "ms_per_call": dt * 1e3, "samples_per_s": B / dt, "ess_ais": info["ess_ais"],
            "log_Z": info["log_Z"]}


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    # SURVEY.md section 8d sizes: 1024 / 16384 (launch-latency bound, D = 32 rows) up to 2^26 (HBM bound, D = 8)
    sizes = (1024, 16384, 1 << 20, 1 << 24) if quick else (1024, 16384, 1 << 20, 1 << 24, 1 << 26)
    res = {"resample": [resample_roofline(n, D=32 if n <= (1 << 20) else 8) for n in sizes]}
    res["ais"] = [
        ais_config("cfg1 GMM-40 2D, RealNVP 4 layers W=80, 512 chains, M=4, Metropolis", 2, 4, 40, 512, 4, "metropolis"),
        ais_config("cfg2 ManyWell-6, RealNVP 8 layers W=240, 1024 chains, M=8, HMC L=5", 6, 8, 40, 1024, 8, "hmc"),
        ais_config("headline ManyWell-32, RealNVP 10 layers W=320, 1024 chains, M=8, HMC L=5", 32, 10, 10, 1024, 8, "hmc"),
        ais_config("cfg3-like ManyWell-32, RealNVP 12 layers W=320, 2048 chains, M=12, HMC L=5", 32, 12, 10, 2048, 12, "hmc"),
        ais_config("ManyWell-32, RealNVP 10 layers, 4096 chains, M=8", 32, 10, 10, 4096, 8, "hmc"),
        ais_config("ManyWell-32, RealNVP 10 layers, 16384 chains, M=8", 32, 10, 10, 16384, 8, "hmc", steps=4),
    ]
    print(json.dumps(res, indent=1))
