"""Lever (b) experiment: does the size of the packed weight image (vs the 4 MiB per-XCD L2) set the time per layer of
k_hmc_step?  Same width (W = 320, D = 32), K = 1..20 layers -> image 1..19 MB; ms per transition and per layer at
B = 1024 (64 workgroups) and B = 16 (ONE workgroup: no L2 sharing between workgroups at all)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from fab_torch_amd.transition_operators import create_point
from fab_torch_amd import _ops

dev = torch.device("cuda", 0)
D, NODES, L = 32, 10, 5
if os.environ.get("FAST") == "1":                    # the same sweep for the bf16 fast mode
    fa.fast_mode(True)
out = []
for K in (1, 2, 3, 4, 6, 10, 16, 20):
    torch.manual_seed(0)
    flow = fa.RealNVP(D, K, NODES).to(dev).requires_grad_(False)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.normal_(0, 0.01); l3.bias.normal_(0, 0.01)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(4, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=L,
                                   eval_mode=True).to(dev)
    row = {"K": K, "image_MB": _ops.load().flow_packed_floats(D, K, D * NODES) * 4 / 1e6}
    for B in (1024, 16):
        x0, _ = flow.native_sample(torch.randn(B, D, device=dev))
        pt = create_point(x0, flow, target, with_grad=True)
        for _ in range(3):
            hmc.transition(pt, 2, 0.4)
        n = 10
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            hmc.transition(pt, 2, 0.4); ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]
        row[f"ms_B{B}"] = ms
        row[f"us_per_layer_pass_B{B}"] = ms * 1e3 / (K * L)          # one forward + one reverse layer
    out.append(row)
    print(json.dumps(row), flush=True)
