"""Dev tool: the fused-stage 8-chain RealNVP kernel (flow_r8f.h) against flow_r8.h - bit identity of a transition and of the
chain initialisation on several shapes, ms per transition of both, stage timeline of the fused kernel."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa
from fab_torch_amd import _ops, _lib
from fab_torch_amd.transition_operators import create_point
dev = torch.device("cuda", 0)
out = {"identity": [], "timing": []}


def run(flow, target, D, B, fused, reps=1):
    with _ops.option(_ops.OPT_TILE_SHAPE, 8), _ops.option(_ops.OPT_R8_FUSED, fused):
        hmc = fa.HamiltonianMonteCarlo(4, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05,
                                       n_outer=2, L=3).to(dev)
        g = torch.Generator(device=dev).manual_seed(5)
        x0, _ = flow.native_sample(torch.randn(B, D, device=dev, generator=g))
        res = []
        for _ in range(reps):
            pt = create_point(x0.clone(), flow, target, with_grad=True)
            torch.manual_seed(77)
            lw = torch.zeros(B, device=dev)
            hmc.transition(pt, 2, 0.4, log_w=lw, beta_next=0.6)
            res.append([t.clone() for t in (pt.x, pt.log_q, pt.log_p, pt.grad_log_q, pt.grad_log_p, lw, hmc.epsilons.clone())])
        return res


for (D, K, nodes, B) in [(32, 10, 10, 2048), (32, 10, 8, 2048), (32, 3, 9, 100), (16, 3, 16, 257), (8, 2, 32, 29), (20, 2, 13, 50),
                         (6, 2, 40, 33), (32, 12, 8, 1501), (5, 1, 64, 17)]:
    torch.manual_seed(D * 100 + K)
    flow = fa.RealNVP(D, K, nodes).to(dev).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    if D % 2:
        target = fa.GMM(D, n_mixes=5, loc_scaling=2.0, seed=1, true_expectation_estimation_n_samples=1000) if D == 2 else None
    else:
        target = fa.ManyWellEnergy(D)
    if target is None:
        continue
    a = run(flow, target, D, B, 0)[0]
    b = run(flow, target, D, B, 1, reps=4)
    same = all(torch.equal(x, y) for x, y in zip(a, b[0]))
    det = all(all(torch.equal(x, y) for x, y in zip(b[0], r)) for r in b[1:])
    md = max(float((x - y).abs().max()) for x, y in zip(a, b[0]))
    out["identity"].append({"D": D, "K": K, "W": nodes * D, "B": B, "bit_identical": same, "deterministic": det, "max_abs_diff": md,
                            "finite": bool(all(torch.isfinite(x).all() for x in b[0][:2]))})
    print(out["identity"][-1], flush=True)

# ms per transition at the headline architecture / hidden width 256
for nodes in (10, 8):
    D = 32
    torch.manual_seed(0)
    flow = fa.RealNVP(D, 10, nodes).to(dev).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(8, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=5).to(dev)
    hmc.set_eval_mode(True)
    for fused in (0, 1):
        row = {}
        for B in (1024, 2048):
            with _ops.option(_ops.OPT_TILE_SHAPE, 8), _ops.option(_ops.OPT_R8_FUSED, fused):
                x0, _ = flow.native_sample(torch.randn(B, D, device=dev))
                pt = create_point(x0, flow, target, with_grad=True)
                for _ in range(5):
                    hmc.transition(pt, 4, 0.5)
                n = 20
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
                ev[0].record()
                for i in range(n):
                    hmc.transition(pt, 4, 0.5)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                row[B] = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]
        out["timing"].append({"W": nodes * D, "fused": fused, "ms_per_transition": row})
        print(out["timing"][-1], flush=True)

# stage timeline of the fused kernel (workgroup 0)
_ops.load().set_option(_ops.OPT_TIMELINE, 1)
for nodes in (10, 8):
    D, B = 32, 2048
    torch.manual_seed(0)
    flow = fa.RealNVP(D, 10, nodes).to(dev).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(8, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, L=5).to(dev)
    hmc.set_eval_mode(True)
    with _ops.option(_ops.OPT_TILE_SHAPE, 8), _ops.option(_ops.OPT_R8_FUSED, 1):
        x0, _ = flow.native_sample(torch.randn(B, D, device=dev))
        pt = create_point(x0, flow, target, with_grad=True)
        for _ in range(5):
            hmc.transition(pt, 4, 0.5)
        torch.cuda.synchronize()
    buf = (C.c_int64 * 64)()
    _lib.check(_lib.load().fabhip_debug_timeline(buf, 64), "timeline")
    ts = list(buf)
    names = {0: "fwd layer start", 1: "F1: D x D map, W1", 2: "F2: W2", 3: "F3: W3 K-split, next ring", 4: "coupling (every wave)",
             16: "rev layer start", 17: "R1: W3T", 18: "R2: W2T", 19: "R3: W1T K-split", 20: "sum, D x D map^T, next ring"}
    print(f"timeline W={nodes * D}")
    prev = None
    for i in sorted(names):
        if not ts[i]:
            continue
        print(f"{i:2d} {names[i]:36s}", "" if prev is None or i in (0, 16) else f"+{ts[i] - prev:6d} ticks")
        prev = ts[i]
    out[f"timeline_W{nodes * D}"] = {names[i]: ts[i] - ts[0 if i < 16 else 16] for i in names if ts[i]}
_ops.load().set_option(_ops.OPT_TIMELINE, 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/dev_r8f.json", "w"), indent=1)
