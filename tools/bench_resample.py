"""Resample pipeline at N = 2^26 (HBM roofline size): per-stage HIP-event timings through the custom ops."""
import json, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fab_torch_amd as fa
from fab_torch_amd import _ops

def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ms[n // 2] * 1e-3

def main():
    ops = _ops.load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
    g = torch.Generator(device="cuda").manual_seed(0)
    lw = torch.randn(N, device="cuda", generator=g) * 3
    out = {"N": N}
    ws = ops.fixed_cdf(lw, None)
    t = timeit(lambda: ops.fixed_cdf(lw, ws))
    out["scan_only"] = {"s": t, "alg_bytes": 12 * N, "TBps": 12 * N / t / 1e12}
    t = timeit(lambda: ops.fixed_cdf(lw, None))
    out["max_plus_scan"] = {"s": t, "alg_bytes": 12 * N, "TBps": 12 * N / t / 1e12}
    t = timeit(lambda: ops.resample_systematic(lw, 0.3, N))
    out["systematic_fused_e2e"] = {"s": t, "alg_bytes": 12 * N, "TBps": 12 * N / t / 1e12, "traffic_bytes": 20 * N,
                                   "traffic_TBps": 20 * N / t / 1e12}
    _opt = _ops.option(_ops.OPT_SYSTEMATIC_VARIANT, 0)
    t = timeit(lambda: ops.resample_systematic(lw, 0.3, N))
    _opt.__exit__()
    out["systematic_cdf_in_hbm_e2e"] = {"s": t, "alg_bytes": 12 * N, "TBps": 12 * N / t / 1e12}
    t = timeit(lambda: ops.ess_logz(lw, None, float(N)))
    out["ess_logz"] = {"s": t, "alg_bytes": 4 * N, "TBps": 4 * N / t / 1e12}
    print(json.dumps(out))

main()
