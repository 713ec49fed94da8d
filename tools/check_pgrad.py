"""Dev check: stream-K parameter gradients (FABHIP_PGRAD=1) against the round-1 block kernel (0) on the same tape, several shapes;
then timings at the trainer's minibatch shape.  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _ops
dev = torch.device("cuda", 0)
ops = _ops.load()
SHAPES = [] if "--bench-only" in sys.argv else [(32, 10, 10, 2048, False), (6, 8, 40, 1000, False), (2, 4, 40, 77, False), (60, 3, 4, 333, False),
                             (33, 2, 8, 130, True), (12, 3, 20, 4096, False), (32, 2, 6, 5, False)]
for (D, K, nodes, B, an) in SHAPES:
    torch.manual_seed(0)
    flow = fa.make_wrapped_normflow_realnvp(D, K, nodes, act_norm=an).to(dev)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.add_(0.01 * torch.randn_like(l3.weight)); l3.bias.add_(0.01 * torch.randn_like(l3.bias))
    x = torch.randn(B, D, device=dev)
    coef = torch.randn(B, device=dev) / B
    lq, tape = flow.log_prob_with_tape(x)
    res = {}
    for mode in (0, 1):
        with _ops.option(_ops.OPT_PGRAD, mode):
            res[mode] = [flow.param_grad_flat(tape, coef).clone() for _ in range(3)]
    a, b = res[0][0], res[1][0]
    det = all(torch.equal(b, r) for r in res[1])
    scale = a.abs().max().item()
    err = (a - b).abs().max().item()
    assert bool(torch.isfinite(lq).all())
    print(f"D={D} K={K} W={D*nodes} B={B} an={an}: max|old-new| = {err:.3e} (scale {scale:.3e}, rel {err/scale:.2e})  deterministic={det}  finite={bool(torch.isfinite(b).all())}")
    assert err <= 2e-5 * scale + 1e-7 and det, "MISMATCH"

def ev(fn, n=30):
    for _ in range(5): fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        fn(); e[i + 1].record()
    torch.cuda.synchronize()
    return sorted(e[i].elapsed_time(e[i + 1]) for i in range(n))[n // 2] * 1e3
torch.manual_seed(0)
flow = fa.make_wrapped_normflow_realnvp(32, 10, 10, act_norm=False).to(dev)
x = torch.randn(2048, 32, device=dev); coef = torch.randn(2048, device=dev) / 2048
lq, tape = flow.log_prob_with_tape(x)
for mode in (0, 1):
    with _ops.option(_ops.OPT_PGRAD, mode):
        print(f"FABHIP_PGRAD={mode}: param_grad_flat {ev(lambda: flow.param_grad_flat(tape, coef)):.1f} us (incl. affine grads, events)")
