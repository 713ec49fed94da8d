import os, sys, copy, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import fab_torch_amd as fa
from oracle import flow as oflow
from test_gpu_fast_mode import emulation
D,K,nodes=32,1,10
B,M=8,1
W=D*nodes
for pat in ('evenq','oddq','k0_79','k80_159','n0_63','n256_319','lane0_31'):
    torch.manual_seed(1)
    nf = oflow.make_realnvp(D, K, nodes); oflow.randomize_last_layers(nf, 0.02, 5)
    with torch.no_grad():
        for f in nf.flows:
            if isinstance(f, oflow.AffineCouplingBlock):
                l2 = f.flows[1].param_map.net[2]
                w = torch.zeros(W, W)
                if pat == 'diag': w = 0.5*torch.eye(W)
                elif pat.startswith('shift'):
                    s = int(pat[5:]); idx = torch.arange(W); w[idx, (idx+s) % W] = 0.5
                elif pat == 'rand_block0': w[:64, :] = 0.05*torch.randn(64, W)
                elif pat == 'rand_k80': w[:, :80] = 0.05*torch.randn(W, 80)
                else:
                    w = l2.weight.detach().clone()
                    kk = torch.arange(W)
                    if pat == 'evenq': w[:, (kk % 8) >= 4] = 0
                    elif pat == 'oddq': w[:, (kk % 8) < 4] = 0
                    elif pat == 'k0_79': w[:, 80:] = 0
                    elif pat == 'k80_159': w[:, :80] = 0; w[:, 160:] = 0
                    elif pat == 'n0_63': w[64:, :] = 0
                    elif pat == 'n256_319': w[:256, :] = 0
                    elif pat == 'lane0_31': w[(kk % 64) >= 32, :] = 0
                l2.weight.copy_(w)
    hf = fa.RealNVP(D, K, nodes); hf._nf_model.load_state_dict(nf.state_dict()); hf = hf.to('cuda').requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    g = torch.Generator(device='cuda').manual_seed(3)
    eps0 = torch.randn(B, D, device='cuda', generator=g); na = torch.randn(M,1,B,D, device='cuda', generator=g); nb = 1e9*torch.ones(M,1,B, device='cuda')
    hmc = fa.HamiltonianMonteCarlo(M, D, hf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=1e-6, L=1, eval_mode=True).to('cuda')
    ais = fa.AnnealedImportanceSampler(hf, target.log_prob, hmc, False, 2.0, M)
    with fa.fast_mode(pat != 'default_fp32'):
        pt, log_w, n_valid, stats, bx, blw = ais.run(B, eps0, na, nb, want_base=True)
    em = emulation(nf)
    nv = int(n_valid[1])
    lq_e = em.log_prob(pt.x[:nv].cpu().double()).detach()
    print(pat, 'per chain', [round(float(v),5) for v in (pt.log_q[:nv].cpu().double() - lq_e).abs()], 'log_p', [round(float(v),3) for v in pt.log_p[:4].cpu()])
    print(pat, 'n_valid', n_valid.cpu().tolist(), 'max |lq - em|', float((pt.log_q.cpu().double()[:int(n_valid[1])] - lq_e[:int(n_valid[1])]).abs().max()) if int(n_valid[1]) else None, pt.log_q[:2].cpu().tolist(), lq_e[:2].tolist())
