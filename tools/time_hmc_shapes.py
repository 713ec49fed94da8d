import os, sys, torch
sys.path.insert(0, "/root/repo")
import fab_torch_amd as fa
from fab_torch_amd import _ops
from fab_torch_amd.transition_operators import create_point
dev = torch.device("cuda", 0)
D = 32
for nodes in (8, 10):
    torch.manual_seed(0)
    flow = fa.RealNVP(D, 10, nodes).to(dev).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(8, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=5).to(dev)
    hmc.set_eval_mode(True)
    for shape in (4, 8, 16):
        row = []
        for B in (256, 512, 1024, 1536, 2048):
            with _ops.option(_ops.OPT_TILE_SHAPE, shape):
                x0, _ = flow.native_sample(torch.randn(B, D, device=dev))
                pt = create_point(x0, flow, target, with_grad=True)
                for _ in range(3):
                    hmc.transition(pt, 4, 0.5)
                n = 10
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
                ev[0].record()
                for i in range(n):
                    hmc.transition(pt, 4, 0.5)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                row.append(sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2])
        print(f"W={nodes*D} tile {shape:2d}: " + "  ".join(f"{v:.3f}" for v in row), "(ms per transition at 256 512 1024 1536 2048 chains)")
