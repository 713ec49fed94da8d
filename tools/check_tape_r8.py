"""Dev check: tape forward on 8-chain tiles (default) against the 16-chain kernel (FABHIP_TAPE_TILES=16): log q, d/dx, the tape's
matrices and the parameter gradients; timings at the trainer's minibatch shape.  GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd import _ops
dev = torch.device("cuda", 0)
ops = _ops.load()
SHAPES = [] if "--bench-only" in sys.argv else [(32, 10, 10, 2048), (32, 3, 8, 100), (6, 8, 40, 1000), (12, 3, 20, 37), (32, 2, 10, 8)]
for (D, K, nodes, B) in SHAPES:
    torch.manual_seed(0)
    flow = fa.make_wrapped_normflow_realnvp(D, K, nodes, act_norm=False).to(dev)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.add_(0.01 * torch.randn_like(l3.weight)); l3.bias.add_(0.01 * torch.randn_like(l3.bias))
    x = torch.randn(B, D, device=dev)
    coef = torch.randn(B, device=dev) / B
    out = {}
    for mode in (16, 0):
        with _ops.option(_ops.OPT_TAPE_TILES, mode):
            lq, tape, gx = flow.log_prob_with_tape(x, want_grad_x=True)
            g = flow.param_grad_flat(tape, coef)
            lq2, tape2, gx2 = flow.log_prob_with_tape(x, want_grad_x=True)
            out[mode] = (lq.clone(), gx.clone(), g.clone(), tape[0].clone(), (lq2.clone(), tape2[0].clone()))
    a, b = out[16], out[0]
    lay = [int(v) for v in ops.flow_tape_layout(D, K, D * nodes, B)]
    total = lay[17]
    det = torch.equal(b[0], b[4][0])      # (the tape holds unwritten padding: TB columns D .. wz)
    rel = lambda u, v: ((u - v).abs().max() / v.abs().max().clamp(min=1e-30)).item()
    print(f"D={D} K={K} W={D*nodes} B={B}: log_q rel {rel(b[0], a[0]):.2e}  grad_x rel {rel(b[1], a[1]):.2e}  param grads rel {rel(b[2], a[2]):.2e}  "
          f"tape rel {rel(b[3][:total], a[3][:total]):.2e}  deterministic={det}")
    assert rel(b[0], a[0]) < 1e-5 and det

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
x = torch.randn(2048, 32, device=dev)
for mode in (16, 0):
    with _ops.option(_ops.OPT_TAPE_TILES, mode):
        print(f"FABHIP_TAPE_TILES={mode}: log_prob_with_tape {ev(lambda: flow.log_prob_with_tape(x)):.1f} us (events, incl. the tape allocation)")
