"""Soak / race screen of the stream kernels (hand-counted s_waitcnt, LDS-only barriers): the same call repeated many times
must return the same bits.  RealNVP AIS calls on 4-chain (fused stages: AGPR ring) and 8-chain tiles - with the in-kernel
step-size rule (FABHIP_OPT_ADAPT_FOLD: ordered by device-scope relaxed atomics + a ticket) AND against the separate
k_hmc_adapt launch: outputs and adapted step sizes must agree bit for bit on every repetition - in fast mode on the 4-chain
tiles, and spline density + gradient evaluations on 4 / 8 / 16-chain tiles; several batch sizes, REPS repetitions each
(default 40)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fab_torch_amd as fa
from fab_torch_amd import _ops

DEV, REPS = "cuda", int(os.environ.get("REPS", 40))
bad = 0
D, M = 32, 4
for nodes in (10, 8):
    torch.manual_seed(1)
    flow = fa.RealNVP(D, 10, nodes).to(DEV).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    for shape, sizes in ((8, (2048, 1499, 517, 8)), (4, (1024, 1027, 517, 3))):
        for B in sizes:
            g = torch.Generator().manual_seed(3)
            eps0 = torch.randn(B, D, generator=g).to(DEV)
            na = torch.randn(M, 1, B, D, generator=g).to(DEV)
            nb = torch.empty(M, 1, B).exponential_(generator=g).to(DEV)
            ref = None
            with _ops.option(_ops.OPT_TILE_SHAPE, shape):
                for rep in range(REPS):
                    # odd repetitions: the step-size rule as its own launch (k_hmc_adapt) - the same sums in the same order
                    with _ops.option(_ops.OPT_ADAPT_FOLD, rep % 2 == 0):
                        hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=5).to(DEV)
                        ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
                        for call in range(2):              # the second call starts from the first one's adapted step sizes
                            pt, lw = ais.sample_and_log_weights(B, eps0=eps0, noise_a=na, noise_b=nb)
                    cur = (pt.x.clone(), lw.clone(), hmc.epsilons.clone(), hmc.common_epsilon.clone())
                    if ref is None:
                        ref = cur
                    elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                        bad += 1
            print(f"realnvp W={nodes * D} {shape}-chain tiles B={B}: {REPS} repetitions (rule in the kernel / own launch alternating), "
                  f"mismatches so far {bad}", flush=True)
    for B in (1024, 37):                                   # fast mode on the 4-chain tiles (bf16 W x W tiles)
        g = torch.Generator().manual_seed(5)
        eps0 = torch.randn(B, D, generator=g).to(DEV)
        na = torch.randn(M, 1, B, D, generator=g).to(DEV)
        nb = torch.empty(M, 1, B).exponential_(generator=g).to(DEV)
        ref = None
        for rep in range(REPS):
            hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=5).to(DEV)
            ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
            with fa.fast_mode():
                pt, lw = ais.sample_and_log_weights(B, eps0=eps0, noise_a=na, noise_b=nb)
            cur = (pt.x.clone(), lw.clone(), hmc.epsilons.clone())
            if ref is None:
                ref = cur
            elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                bad += 1
        print(f"realnvp W={nodes * D} fast mode B={B}: {REPS} repetitions, mismatches so far {bad}", flush=True)
for Dd, H, L in ((32, 256, 12), (60, 256, 6), (8, 200, 3)):
    torch.manual_seed(2)
    hf = fa.make_wrapped_normflow_spline(Dd, L, H, (), 5.0).to(DEV).requires_grad_(False)
    with torch.no_grad():
        for p in hf.parameters():
            if p.dim() == 2:
                p.add_(0.03 * torch.randn_like(p))
    for B in (2048, 1027, 130, 3):
        x = (1.5 * torch.randn(B, Dd, generator=torch.Generator().manual_seed(B))).to(DEV)
        for shape in (4, 8, 16):
            ref = None
            with _ops.option(_ops.OPT_TILE_SHAPE, shape):
                for rep in range(REPS):
                    cur = hf.log_prob_and_grad(x)
                    if ref is None:
                        ref = (cur[0].clone(), cur[1].clone())
                    elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])):
                        bad += 1
        print(f"spline D={Dd} H={H} L={L} B={B}: 3 tile shapes x {REPS} repetitions, mismatches so far {bad}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
