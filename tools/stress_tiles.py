"""Randomised agreement sweep of the tile shapes: the fused RealNVP HMC transition (16 / 8 / 4 chains per workgroup) and the
spline density kernels (16x16x4; 4x4x1 with 4 / 8 / 16 chains) on random dimensions, depths, widths and ragged batches.
Prints one line per case; exits non-zero on a disagreement beyond rounding (ReLU- / accept-flips excepted as in the tests)."""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa
from fab_torch_amd import _ops

DEV = "cuda"
rng = random.Random(int(os.environ.get("SEED", 0)))
bad = 0
for case in range(int(os.environ.get("N", 24))):
    D = rng.choice([2, 4, 6, 8, 10, 16, 20, 26, 32])
    W = rng.choice([200, 240, 256, 260, 300, 320])
    nodes_w = max(1, W // D)
    K = rng.randint(1, 5)
    B = rng.choice([1, 7, 8, 9, 33, 100, 257, 1000, 1500, 2048])
    torch.manual_seed(case)
    flow = fa.RealNVP(D, K, nodes_w).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] != p.shape[1]:
                p.add_(0.05 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D) if D >= 4 else fa.GMM(D, n_mixes=5, loc_scaling=2.0, seed=1, true_expectation_estimation_n_samples=100)
    res = {}
    for shape in (16, 8, 4):
        with _ops.option(_ops.OPT_TILE_SHAPE, shape):
            hmc = fa.HamiltonianMonteCarlo(3, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.05, n_outer=1, L=3,
                                           eval_mode=True).to(DEV)
            g = torch.Generator(device=DEV).manual_seed(5)
            x0, _ = flow.native_sample(torch.randn(B, D, device=DEV, generator=g))
            pt = fa.create_point(x0, flow, target, with_grad=True)
            torch.manual_seed(77)
            out = hmc.transition(pt, 2, 0.4)
            res[shape] = (out.x.clone(), out.log_q.clone())
    line = f"realnvp D={D} W={nodes_w * D} K={K} B={B}:"
    for shape in (8, 4):
        same = (res[16][0] - res[shape][0]).abs().amax(dim=1) <= 1e-4 * (1 + res[16][0].abs().amax(dim=1))
        frac = float(same.float().mean())
        dl = float((res[16][1][same] - res[shape][1][same]).abs().max()) if same.any() else 0.0
        used = not torch.equal(res[16][1], res[shape][1])
        ok = frac >= 0.97 and dl <= 2e-3 * (1 + float(res[16][1].abs().max()))
        bad += not ok
        line += f"  tile {shape}: {'own kernel' if used else 'fell back'} agree {frac:.3f} dlogq {dl:.1e} {'ok' if ok else 'BAD'}"
    print(line, flush=True)
for case in range(int(os.environ.get("NS", 12))):
    D = rng.choice([2, 5, 8, 13, 21, 32, 40, 60, 64])
    H = rng.choice([200, 225, 256])
    L = rng.randint(1, 6)
    B = rng.choice([1, 3, 4, 5, 31, 130, 1027, 2049])
    circ = tuple(sorted(rng.sample(range(D), rng.randint(0, min(4, D)))))
    tb = torch.full((D,), 5.0); tb[list(circ)] = math.pi
    torch.manual_seed(100 + case)
    hf = fa.make_wrapped_normflow_spline(D, L, H, circ, tb).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in hf.parameters():
            if p.dim() == 2:
                p.add_(0.03 * torch.randn_like(p))
    x = (1.5 * torch.randn(B, D, generator=torch.Generator().manual_seed(case))).to(DEV)
    out = {}
    for name, mfma, shape in (("16x16x4", 16, 0), ("r16", 0, 16), ("r8", 0, 8), ("r4", 0, 4)):
        with _ops.option(_ops.OPT_SPLINE_MFMA, mfma), _ops.option(_ops.OPT_TILE_SHAPE, shape):
            out[name] = hf.log_prob_and_grad(x)
    bit = all(torch.equal(out["r8"][i], out[k][i]) for k in ("r16", "r4") for i in (0, 1))
    dl = float((out["r8"][0] - out["16x16x4"][0]).abs().max() / (1 + out["16x16x4"][0].abs().max()))
    rel = (out["r8"][1] - out["16x16x4"][1]).norm(dim=1) / out["16x16x4"][1].norm(dim=1).clamp_min(1e-6)
    ok = bit and dl < 1e-4 and float(rel.median()) < 1e-4 and bool(torch.isfinite(out["r8"][0]).all())
    bad += not ok
    print(f"spline D={D} H={H} L={L} B={B} circ={len(circ)}: stream tiles bit-identical {bit}, vs 16x16x4 dlogq {dl:.1e} "
          f"grad median {float(rel.median()):.1e} worst {float(rel.max()):.1e} {'ok' if ok else 'BAD'}", flush=True)
print("disagreements:", bad)
sys.exit(1 if bad else 0)
