"""Micro-benchmark of the flow kernels: per-layer cost of log_prob (fwd) and log_prob+grad (fwd+bwd) at
different hidden widths -> separates fixed per-stage overhead from MFMA time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa

dev = torch.device("cuda", 0)
def ev(fn, n=20, warm=5):
    for _ in range(warm): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        fn(); e[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(e[i].elapsed_time(e[i + 1]) for i in range(n))
    return ms[n // 2] * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for nodes in (2, 4, 10, 16):
    K = 10
    torch.manual_seed(0)
    flow = fa.RealNVP(32, K, nodes).to(dev).requires_grad_(False)
    x = torch.randn(B, 32, device=dev)
    t_f = ev(lambda: flow.native_log_prob(x, with_grad=False))
    t_g = ev(lambda: flow.native_log_prob(x, with_grad=True))
    print(f"W={32*nodes:4d} B={B}: fwd {t_f:8.1f} us ({t_f/K:6.2f}/layer)   fwd+bwd {t_g:8.1f} us ({t_g/K:6.2f}/layer, bwd {(t_g-t_f)/K:6.2f}/layer)")
