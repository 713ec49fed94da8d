import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import fab_torch_amd as fa
from fab_torch_amd import _ops
from helpers import load_golden, flow_from_g14
from test_gpu_parity import hip_flow_from_oracle
DEV='cuda'
g = load_golden("g16_ais_headline_rejecting.npz")
nf = flow_from_g14(g); hf = hip_flow_from_oracle(nf)
D, M, B, L, alpha = int(g["D"]), int(g["M"]), g["eps0"].shape[0], int(g["L"]), float(g["alpha"])
target = fa.ManyWellEnergy(D)
T = lambda k: torch.tensor(g[k]).to(DEV)
betas = torch.tensor(g["B_space"])
for shape in (4, 8, 16):
    with _ops.option(_ops.OPT_TILE_SHAPE, shape):
        hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=alpha, p_target=False, L=L, eval_mode=True).to(DEV)
        hmc.epsilons.copy_(T("in_epsilons")); hmc.common_epsilon.copy_(T("in_common_epsilon"))
        pt = fa.create_point(T("snap_x")[0].clone(), hf, target, with_grad=True)
        for j in range(1, M + 1):
            pt = hmc.transition(pt, j, float(betas[j]), noise_p=T("noise_p")[j - 1], noise_e=T("noise_e")[j - 1])
            err = (pt.x.cpu() - torch.tensor(g["snap_x"][j])).abs().max(1).values
            k = int(err.argmax())
            print(f"shape {shape} free-running after transition {j}: max x err {float(err.max()):.2e} (chain {k}); chains > 1e-4: {int((err > 1e-4).sum())}")
        ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, alpha, M)
        p2, lw = ais.sample_and_log_weights(B, eps0=T("eps0"), noise_a=T("noise_p"), noise_b=T("noise_e"))
        err = (p2.x.cpu() - torch.tensor(g["out_x"])).abs().max(1).values
        print(f"shape {shape} fused: max x err {float(err.max()):.2e} chain {int(err.argmax())}")
print("---- x0 = flow.sample(eps0) against the reference's starting state")
for shape in (4, 16):
    with _ops.option(_ops.OPT_TILE_SHAPE, shape):
        x0, lq0 = hf.native_sample(T("eps0"))
        e = (x0.cpu() - torch.tensor(g["snap_x"][0])).abs().max(1).values
        print(f"shape {shape}: max |x0 - ref| {float(e.max()):.2e} chain {int(e.argmax())}; lq0 err {float((lq0.cpu() - torch.tensor(g['snap_log_q'][0])).abs().max()):.2e}")
        # fused call with M' = 1 .. : where does the deviation appear
        for Mj in (1, 2, 4, 8):
            hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=alpha, p_target=False, L=L, eval_mode=True).to(DEV)
            hmc.epsilons.copy_(T("in_epsilons")); hmc.common_epsilon.copy_(T("in_common_epsilon"))
            pt = fa.create_point(x0.clone(), hf, target, with_grad=True)
            for j in range(1, Mj + 1):
                pt = hmc.transition(pt, j, float(betas[j]), noise_p=T("noise_p")[j - 1], noise_e=T("noise_e")[j - 1])
            e = (pt.x.cpu() - torch.tensor(g["snap_x"][Mj])).abs().max(1).values
            print(f"   from the HIP x0, {Mj} transitions step by step: max err {float(e.max()):.2e} chain {int(e.argmax())}")
