"""Generate golden fixtures by RUNNING THE IMPORTED REFERENCE (fab-torch) in this container.

Usage (build container only — /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

The reference's AIS / HMC / Metropolis / Point / targets / ESS / resample code is imported
from /root/reference (with the three absent third-party modules stubbed)
and driven with the oracle RealNVP (oracle/flow.py) as its ``base_distribution`` plug-in —
the reference only needs the `Distribution` interface (fab/sampling_methods/ais.py:23,56-62).
All noise the reference draws is captured by wrapping torch's RNG entry points, so that the
fixtures hold (inputs, noise, outputs) triples.  Only DATA is written: small .npz files.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

for name in ["wandb", "normflows", "nflows", "nflows.flows"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["normflows"].NormalizingFlow = object
sys.modules["nflows"].flows = sys.modules["nflows.flows"]
sys.modules["nflows.flows"].Flow = object
sys.path.insert(0, "/root/reference")

from fab import AnnealedImportanceSampler, HamiltonianMonteCarlo, Metropolis  # noqa: E402
from fab.sampling_methods.base import (Point, get_intermediate_log_prob,  # noqa: E402
                                       get_grad_intermediate_log_prob, resample, create_point)
from fab.target_distributions.many_well import ManyWellEnergy  # noqa: E402
from fab.target_distributions.gmm import GMM  # noqa: E402
from fab.utils.numerical import effective_sample_size  # noqa: E402

from oracle import flow as oflow  # noqa: E402


def npz(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in kw.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def flow_state(nf):
    return {"flow." + k: v for k, v in nf.state_dict().items()}


class Capture:
    """Wrap the RNG entry points the reference uses and record every draw."""

    def __init__(self):
        self.randn_like, self.expo, self.randn, self.rand = [], [], [], []

    def __enter__(self):
        self._o = (torch.randn_like, torch.distributions.Exponential.sample, torch.randn, torch.rand)
        o_rl, o_ex, o_rn, o_ra = self._o
        cap = self

        def randn_like(x, *a, **k):
            t = o_rl(x, *a, **k); cap.randn_like.append(t.clone()); return t

        def expo(self_, shape=torch.Size()):
            t = o_ex(self_, shape); cap.expo.append(t.clone()); return t

        def randn(*a, **k):
            t = o_rn(*a, **k); cap.randn.append(t.clone()); return t

        def rand(*a, **k):
            t = o_ra(*a, **k); cap.rand.append(t.clone()); return t

        torch.randn_like = randn_like
        torch.distributions.Exponential.sample = expo
        torch.randn = randn
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.distributions.Exponential.sample, torch.randn, torch.rand = self._o


class Inject:
    """The reverse of Capture: the reference's next `torch.randn_like` / `Exponential.sample` calls return the given draws, in
    order (a re-run of selected chains on the noise a larger run captured for them)."""

    def __init__(self, randn_like, expo):
        self.rl, self.ex = list(randn_like), list(expo)

    def __enter__(self):
        self._o = (torch.randn_like, torch.distributions.Exponential.sample)
        inj = self

        def randn_like(x, *a, **k):
            t = inj.rl.pop(0); assert t.shape == x.shape; return t.clone()

        def expo(self_, shape=torch.Size()):
            t = inj.ex.pop(0); assert tuple(t.shape) == tuple(shape); return t.clone()

        torch.randn_like = randn_like
        torch.distributions.Exponential.sample = expo
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.distributions.Exponential.sample = self._o


class EpsFlow:
    """`Distribution` adapter: deterministic base noise for the oracle flow."""

    def __init__(self, nf, eps):
        self.nf, self.eps = nf, eps

    def sample_and_log_prob(self, shape):
        assert shape[0] == self.eps.shape[0]
        with torch.no_grad():
            return self.nf.sample_eps(self.eps)

    def log_prob(self, x):
        return self.nf.log_prob(x)

    def sample(self, shape):
        return self.sample_and_log_prob(shape)[0]

    @property
    def event_shape(self):
        return self.nf.q0.shape


def make_flow(dim, n_layers, nodes, seed):
    torch.manual_seed(seed)
    nf = oflow.make_realnvp(dim, n_layers, nodes)
    oflow.randomize_last_layers(nf, std=0.05, seed=seed + 1)
    return nf


def g1_beta():
    out = {}
    for M in (1, 4, 8, 12):
        for sp in ("linear", "geometric"):
            ais = AnnealedImportanceSampler.__new__(AnnealedImportanceSampler)
            ais.n_intermediate_distributions = M
            out[f"{sp}_{M}"] = ais.setup_distribution_spacing(sp, M)
    npz("g1_beta.npz", **out)


def g2_intermediate():
    torch.manual_seed(11)
    B, D = 64, 6
    pt = Point(torch.randn(B, D), torch.randn(B) * 3, torch.randn(B) * 5, torch.randn(B, D), torch.randn(B, D) * 4)
    betas = torch.tensor(np.linspace(0, 1, 6))
    out = dict(x=pt.x, log_q=pt.log_q, log_p=pt.log_p, grad_log_q=pt.grad_log_q, grad_log_p=pt.grad_log_p,
               betas=betas)
    for ai, alpha in enumerate((2.0, 0.5)):
        for p_target in (False, True):
            for bi, beta in enumerate(betas):
                out[f"lp_a{ai}_p{int(p_target)}_b{bi}"] = get_intermediate_log_prob(pt, beta, alpha, p_target)
                out[f"gr_a{ai}_p{int(p_target)}_b{bi}"] = get_grad_intermediate_log_prob(pt, beta, alpha, p_target)
    npz("g2_intermediate.npz", **out)


def g3_targets():
    torch.manual_seed(12)
    out = {}
    for D in (6, 32):
        t = ManyWellEnergy(dim=D, use_gpu=False)
        x = torch.randn(64, D) * 1.5
        x[0] = 1.7
        x[1, 0::2] = 1.7; x[1, 1::2] = 0.0
        x[2] = float("nan"); x[3, 0] = float("inf"); x[4] = 40.0
        xg = x.clone().requires_grad_(True)
        lp = t.log_prob(xg)
        g = torch.autograd.grad(lp, xg, torch.ones_like(lp))[0]
        out[f"mw{D}_x"], out[f"mw{D}_lp"], out[f"mw{D}_g"] = x, lp, g
        out[f"mw{D}_logZ"] = t.log_Z
    torch.manual_seed(0)
    gmm = GMM(dim=2, n_mixes=40, loc_scaling=40.0, log_var_scaling=1.0, use_gpu=False,
              true_expectation_estimation_n_samples=1000)
    torch.manual_seed(13)
    x = torch.randn(96, 2) * 30
    x[0] = 0.0
    x[1] = 1e4                      # triggers the -inf mask (log_prob < -1e4)
    x[2] = float("nan")
    out["gmm_locs"] = gmm.locs
    out["gmm_scales"] = torch.diagonal(gmm.scale_trils, dim1=-2, dim2=-1)
    out["gmm_x"], out["gmm_lp"] = x, gmm.log_prob(x)
    npz("g3_targets.npz", **out)


def g4_ess():
    torch.manual_seed(14)
    out = {}
    vecs = [torch.tensor([0., 1., 2., 3.]), torch.randn(64) * 3, torch.randn(1024) * 10 + 150,
            torch.cat([torch.randn(30), torch.tensor([-float("inf")] * 2)]), torch.zeros(17)]
    for i, lw in enumerate(vecs):
        out[f"lw{i}"] = lw
        out[f"ess{i}"] = effective_sample_size(lw)
        lz = torch.logsumexp(lw, dim=0)
        out[f"logZ{i}"] = lz - torch.log(torch.ones_like(lz) * lw.shape[0])
    npz("g4_ess.npz", **out)


def g5_multinomial():
    out = {}
    for N in (64, 1024, 4096, 16384):                           # 16384 = BASELINE cfg 4's gathered particle count
        torch.manual_seed(100 + N)
        lw = torch.randn(N) * 3
        x = torch.arange(N, dtype=torch.float32)[:, None].repeat(1, 2)
        st = torch.get_rng_state()
        xs = resample(x, lw)                                   # the reference's own resample()
        torch.set_rng_state(st)
        u = torch.rand(N, dtype=torch.float64)                 # the uniforms multinomial consumed
        probs = torch.distributions.Categorical(logits=lw).probs
        out[f"lw_{N}"], out[f"u_{N}"], out[f"probs_{N}"] = lw, u, probs
        out[f"idx_{N}"] = xs[:, 0].to(torch.int64)
    npz("g5_multinomial.npz", **out)


def tuned_hmc(M, D, nf, target, eps, L, n_outer, alpha, p_target, tune=True):
    return HamiltonianMonteCarlo(n_ais_intermediate_distributions=M, dim=D, base_log_prob=nf.log_prob,
                                 target_log_prob=target.log_prob, alpha=alpha, p_target=p_target,
                                 epsilon=eps, n_outer=n_outer, L=L, eval_mode=not tune)


def g6_hmc():
    for tag, D, K, nodes, eps, n_outer in (("d6", 6, 3, 5, 0.22, 1), ("d32", 32, 2, 1, 0.22, 1),
                                           ("d6_outer2", 6, 3, 5, 0.22, 2)):
        nf = make_flow(D, K, nodes, seed=20 + D)
        target = ManyWellEnergy(dim=D, use_gpu=False)
        M, L, B, alpha = 4, 5, 64, 2.0
        hmc = tuned_hmc(M, D, nf, target, eps, L, n_outer, alpha, False)
        torch.manual_seed(21)
        with torch.no_grad():
            x0, _ = nf.sample_eps(torch.randn(B, D))
        pt = create_point(x0, nf.log_prob, target.log_prob, with_grad=True)
        i = 2
        beta = torch.tensor(np.linspace(0, 1, M + 2))[i]
        out = dict(beta=beta, i=i, L=L, n_outer=n_outer, alpha=alpha, p_target=0, M=M,
                   in_x=pt.x.clone(), in_log_q=pt.log_q.clone(), in_log_p=pt.log_p.clone(),
                   in_gq=pt.grad_log_q.clone(), in_gp=pt.grad_log_p.clone(),
                   in_epsilons=hmc.epsilons.clone(), in_common_epsilon=hmc.common_epsilon.clone(),
                   mass=hmc.mass_vector.clone())
        with Capture() as cap:
            res = hmc.transition(pt, i, beta)
        out.update(noise_p=torch.stack(cap.randn_like), noise_e=torch.stack(cap.expo),
                   out_x=res.x, out_log_q=res.log_q, out_log_p=res.log_p, out_gq=res.grad_log_q,
                   out_gp=res.grad_log_p, out_epsilons=hmc.epsilons, out_common_epsilon=hmc.common_epsilon,
                   p_accept=torch.stack([v.reshape(()) for v in hmc.first_dist_p_accepts])
                   if i == 1 else torch.zeros(n_outer))
        out.update(flow_state(nf))
        npz(f"g6_hmc_{tag}.npz", **out)


def g7_metropolis():
    D, K, nodes = 2, 2, 8
    nf = make_flow(D, K, nodes, seed=30)
    torch.manual_seed(0)
    target = GMM(dim=2, n_mixes=40, loc_scaling=40.0, log_var_scaling=1.0, use_gpu=False,
                 true_expectation_estimation_n_samples=1000)
    M, B, alpha, n_updates = 4, 64, 2.0, 3
    met = Metropolis(n_ais_intermediate_distributions=M, dim=D, base_log_prob=nf.log_prob,
                     target_log_prob=target.log_prob, n_updates=n_updates, alpha=alpha, p_target=False,
                     max_step_size=5.0, min_step_size=1.0, adjust_step_size=True)
    torch.manual_seed(31)
    with torch.no_grad():
        x0, lq0 = nf.sample_eps(torch.randn(B, D))
        x0 = x0 * 10
    pt = create_point(x0, nf.log_prob, target.log_prob, with_grad=False)
    i = 3
    beta = torch.tensor(np.linspace(0, 1, M + 2))[i]
    out = dict(beta=beta, i=i, n_updates=n_updates, alpha=alpha, p_target=0, M=M,
               in_x=pt.x.clone(), in_log_q=pt.log_q.clone(), in_log_p=pt.log_p.clone(),
               in_noise_scalings=met.noise_scalings.clone(),
               gmm_locs=target.locs, gmm_scales=torch.diagonal(target.scale_trils, dim1=-2, dim2=-1))
    with Capture() as cap:
        res = met.transition(pt, i, beta)
    out.update(noise_x=torch.stack(cap.randn), noise_u=torch.stack(cap.rand),
               out_x=res.x, out_log_q=res.log_q, out_log_p=res.log_p, out_noise_scalings=met.noise_scalings)
    out.update(flow_state(nf))
    npz("g7_metropolis.npz", **out)


def g8_full_chain():
    # (a) ManyWell-6, HMC, M=4, geometric + linear; (b) ManyWell-32, HMC, M=8; (c) GMM-2, Metropolis M=4
    for tag, D, K, nodes, M, spacing, eps, p_target in (
            ("mw6_hmc_m4", 6, 3, 5, 4, "linear", 0.2, False),
            ("mw6_hmc_m8geo_ptarget", 6, 3, 5, 8, "geometric", 0.2, True),
            ("mw32_hmc_m8", 32, 2, 1, 8, "linear", 0.2, False)):
        nf = make_flow(D, K, nodes, seed=40 + D)
        target = ManyWellEnergy(dim=D, use_gpu=False)
        B, L, alpha = 64, 5, 2.0
        hmc = tuned_hmc(M, D, nf, target, eps, L, 1, alpha, p_target)
        torch.manual_seed(41)
        eps0 = torch.randn(B, D)
        ais = AnnealedImportanceSampler(EpsFlow(nf, eps0), target.log_prob, hmc, p_target=p_target,
                                        alpha=alpha, n_intermediate_distributions=M,
                                        distribution_spacing_type=spacing)
        in_eps, in_ceps = hmc.epsilons.clone(), hmc.common_epsilon.clone()
        with Capture() as cap:
            pt, log_w = ais.sample_and_log_weights(B)
        info = ais.get_logging_info()
        out = dict(M=M, L=L, alpha=alpha, p_target=int(p_target), spacing=spacing, eps0=eps0,
                   B_space=ais.B_space, in_epsilons=in_eps, in_common_epsilon=in_ceps,
                   noise_p=torch.stack(cap.randn_like)[:, None], noise_e=torch.stack(cap.expo)[:, None],
                   out_x=pt.x, out_log_q=pt.log_q, out_log_p=pt.log_p, out_gq=pt.grad_log_q,
                   out_gp=pt.grad_log_p, log_w=log_w, out_epsilons=hmc.epsilons,
                   out_common_epsilon=hmc.common_epsilon, ess_base=info["ess_base"],
                   ess_ais=info["ess_ais"], log_Z=info["log_Z"],
                   dist0_p_accept_0=info["dist0_p_accept_0"],
                   average_distance_dist0=info["average_distance_dist0"])
        out.update(flow_state(nf))
        npz(f"g8_ais_{tag}.npz", **out)

    # Metropolis / GMM (cfg 1 shape: D=2, M=4)
    D, K, nodes, M, B, alpha = 2, 2, 8, 4, 64, 2.0
    nf = make_flow(D, K, nodes, seed=50)
    with torch.no_grad():
        nf.q0.log_scale += 2.0         # spread the base over the GMM support
    torch.manual_seed(0)
    target = GMM(dim=2, n_mixes=40, loc_scaling=40.0, log_var_scaling=1.0, use_gpu=False,
                 true_expectation_estimation_n_samples=1000)
    met = Metropolis(n_ais_intermediate_distributions=M, dim=D, base_log_prob=nf.log_prob,
                     target_log_prob=target.log_prob, n_updates=2, alpha=alpha, p_target=False,
                     max_step_size=5.0, min_step_size=2.0, adjust_step_size=True)
    torch.manual_seed(51)
    eps0 = torch.randn(B, D)
    ais = AnnealedImportanceSampler(EpsFlow(nf, eps0), target.log_prob, met, p_target=False, alpha=alpha,
                                    n_intermediate_distributions=M)
    in_ns = met.noise_scalings.clone()
    with Capture() as cap:
        pt, log_w = ais.sample_and_log_weights(B)
    info = ais.get_logging_info()
    out = dict(M=M, n_updates=2, alpha=alpha, p_target=0, eps0=eps0, B_space=ais.B_space,
               in_noise_scalings=in_ns, noise_x=torch.stack(cap.randn).reshape(M, 2, B, D),
               noise_u=torch.stack(cap.rand).reshape(M, 2, B), out_x=pt.x, out_log_q=pt.log_q,
               out_log_p=pt.log_p, log_w=log_w, out_noise_scalings=met.noise_scalings,
               ess_base=info["ess_base"], ess_ais=info["ess_ais"], log_Z=info["log_Z"],
               gmm_locs=target.locs, gmm_scales=torch.diagonal(target.scale_trils, dim1=-2, dim2=-1))
    out.update(flow_state(nf))
    npz("g8_ais_gmm_metropolis.npz", **out)


def g14_headline_arch(name="g14_ais_headline.npz", std=0.05, eps_init=0.2, seed=140, save=True, B=64, tune=True, noise=None):
    """(`g15_ais_headline_mild.npz`: the same call in a MILD regime - last coupling Linears N(0, 0.01^2), initial step size
    0.05 - in which one transition does not amplify fp32 rounding: the GPU test allows no waiver on it)
    The reference's own AIS call and one HMC transition at the HEADLINE flow architecture (many_well.yaml:7-10: RealNVP
    10 x (16-320-320-32) + InvertibleAffine, D = 32; ais.yaml / many_well.yaml:25-29: HMC, L = 5, M = 8 here as in
    BASELINE.json's metric), B = 64 chains.  The 4.8 MB of weights are NOT stored: `helpers.seeded_oracle_flow`
    rebuilds them from the seed (same routine, same torch CPU generator); the fixture holds noise, every transition's
    input step sizes and output state (snapshots for teacher-forced per-transition checks) and the final outputs."""
    from helpers import seeded_oracle_flow
    D, K, nodes, M, L, alpha = 32, 10, 10, 8, 5, 2.0
    nf = seeded_oracle_flow(D, K, nodes, seed, std)
    target = ManyWellEnergy(dim=D, use_gpu=False)
    hmc = tuned_hmc(M, D, nf, target, eps_init, L, 1, alpha, False, tune=tune)
    torch.manual_seed(seed + 1)
    eps0 = torch.randn(B, D) if noise is None else noise[0]      # noise = (eps0, [momenta per transition], [Exp(1) per transition])
    ais = AnnealedImportanceSampler(EpsFlow(nf, eps0), target.log_prob, hmc, p_target=False, alpha=alpha,
                                    n_intermediate_distributions=M)
    in_eps, in_ceps = hmc.epsilons.clone(), hmc.common_epsilon.clone()
    snaps, eps_in, ceps_in = [], [], []
    orig = hmc.transition

    def recording_transition(point, i, beta):
        if i == 1:
            snaps.append((point.x.clone(), point.log_q.clone(), point.log_p.clone()))       # the chains' starting state
        eps_in.append(hmc.epsilons[i - 1].clone()); ceps_in.append(hmc.common_epsilon.clone())
        out = orig(point, i, beta)
        snaps.append((out.x.clone(), out.log_q.clone(), out.log_p.clone()))
        return out
    hmc.transition = recording_transition
    if noise is None:
        with Capture() as cap:
            pt, log_w = ais.sample_and_log_weights(B)
    else:                                              # (the same reference call on given draws)
        cap = Capture()
        cap.randn_like, cap.expo = [t.clone() for t in noise[1]], [t.clone() for t in noise[2]]
        with Inject(noise[1], noise[2]):
            pt, log_w = ais.sample_and_log_weights(eps0.shape[0])
    hmc.transition = orig
    info = ais.get_logging_info()
    out = dict(D=D, K=K, nodes=nodes, flow_seed=seed, flow_std=std, M=M, L=L, alpha=alpha, p_target=0,
        eps0=eps0, B_space=ais.B_space, in_epsilons=in_eps, in_common_epsilon=in_ceps,
        noise_p=torch.stack(cap.randn_like)[:, None], noise_e=torch.stack(cap.expo)[:, None],
        snap_x=torch.stack([s_[0] for s_ in snaps]), snap_log_q=torch.stack([s_[1] for s_ in snaps]),
        snap_log_p=torch.stack([s_[2] for s_ in snaps]),
        tr_epsilon=torch.stack(eps_in), tr_common_epsilon=torch.stack(ceps_in),
        out_x=pt.x, out_log_q=pt.log_q, out_log_p=pt.log_p, out_gq=pt.grad_log_q, out_gp=pt.grad_log_p, log_w=log_w,
        out_epsilons=hmc.epsilons, out_common_epsilon=hmc.common_epsilon, ess_base=info["ess_base"],
        ess_ais=info["ess_ais"], log_Z=info["log_Z"], dist0_p_accept_0=info["dist0_p_accept_0"],
        # a probe of the rebuilt weights: the fixture is only valid for the flow `seeded_oracle_flow` returns
        flow_probe=torch.stack([nf.flows[0].flows[1].param_map.net[2].weight[0, :8].detach(),
                                nf.flows[-2].flows[1].param_map.net[4].weight[1, :8].detach()]))
    if save:
        npz(name, **out)
    return out


def g16_rejecting(name="g16_ais_headline_rejecting.npz", std=0.01, eps_init=0.26, seed=160, pool=512, keep=64, verbose=True):
    """the headline architecture with BOTH accept outcomes and still no ill-conditioned chain.  At a step size at
    which a quarter of the proposals is rejected, some chains of ANY batch sit next to a ReLU kink or an accept threshold, so the
    fixture is a SELECTION: the reference runs `pool` chains (step-size tuning frozen: chains are then independent of each other),
    the float64 oracle - teacher-forced from the reference's snapshots - marks the chains on which, in EVERY transition, the
    reference's fp32 proposal is within 5e-6 (densities: 1e-5) of float64, the accept margin exceeds 0.05 and the decisions agree; `keep` of those
    (rejecting ones first) are re-run through the reference on their own captured noise, and that run is the fixture."""
    import copy
    from helpers import seeded_oracle_flow
    from oracle import ais as oais, targets as otgt
    big = g14_headline_arch("unused", std=std, eps_init=eps_init, seed=seed, save=False, B=pool, tune=False)
    D, M, L, alpha = int(big["D"]), int(big["M"]), int(big["L"]), float(big["alpha"])
    nf64 = copy.deepcopy(seeded_oracle_flow(D, int(big["K"]), int(big["nodes"]), seed, std)).double()
    otarget = otgt.ManyWell(D)
    o64 = oais.HMC(M, D, nf64.log_prob, otarget.log_prob, alpha=alpha, p_target=False, L=L, eval_mode=True, dtype=torch.float64)
    good = torch.ones(pool, dtype=torch.bool)
    rejects = torch.zeros(pool, dtype=torch.long)
    for j in range(1, M + 1):
        o64.epsilons = big["tr_epsilon"].double().clone()
        o64.common_epsilon = big["tr_common_epsilon"][j - 1].double().clone()
        p64 = oais.create_point(big["snap_x"][j - 1].double(), nf64.log_prob, otarget.log_prob, True)
        p64 = o64.transition(p64, j, big["B_space"][j], big["noise_p"][j - 1].double(), big["noise_e"][j - 1].double())
        rx = big["snap_x"][j]
        xs = max(1.0, float(rx.abs().max()))
        dev = (p64.x - rx.double()).abs().max(1).values / xs
        rq, rp = big["snap_log_q"][j].double(), big["snap_log_p"][j].double()
        dq = (p64.log_q - rq).abs() / rq.abs().clamp(min=1.0)
        dp = (p64.log_p - rp).abs() / rp.abs().clamp(min=1.0)
        acc_ref = (rx != big["snap_x"][j - 1]).any(1)
        m = o64.last_margin
        # (an order of magnitude inside the 1e-4 the GPU test allows: another summation order moves a chain as far as the
        #  reference's own fp32 rounding did)
        # margin: 0.05 in log-acceptance - the FREE-RUNNING call drifts by a few 1e-4 in x over eight transitions at this step
        # size, which moves a later transition's log-acceptance by ~1e-2; with 1e-3 (enough for a teacher-forced transition) the
        # fused call flipped a decision on one chain per tile shape
        good &= (dev <= 5e-6) & (dq <= 1e-5) & (dp <= 1e-5) & (m.abs() > 5e-2) & (acc_ref == (m > 0))
        rejects += (~acc_ref).long()
    # ... and along the whole free-running chain: float64 from the same start on the same noise ends where the reference's fp32
    # chain ends (8 transitions x 5 leapfrogs at this step size amplify a rounding difference on some chains)
    o64.epsilons = big["in_epsilons"].double().clone()
    o64.common_epsilon = big["in_common_epsilon"].double().clone()
    a64 = oais.AIS(lambda e: tuple(t.detach() for t in nf64.sample_eps(e)), nf64.log_prob, otarget.log_prob, o64, False, alpha, M)
    p_end, lw_end, _ = a64.sample_and_log_weights(big["eps0"].double(), big["noise_p"].double(), big["noise_e"].double())
    xs = max(1.0, float(big["out_x"].abs().max()))
    good &= ((p_end.x - big["out_x"].double()).abs().max(1).values / xs <= 2e-5)
    good &= ((lw_end - big["log_w"].double()).abs() / big["log_w"].double().abs().clamp(min=1.0) <= 2e-5)
    # ... and the whole chain must not amplify its starting state: the GPU's flow sample differs from the reference's x0 by up to
    # 1e-5 (10 layers of fp32), and ONE transition at this step size amplified that 165-fold on a chain that had passed everything
    # above (random perturbations do not find such a direction in 32 dimensions).  Finite-difference Jacobian of the final state
    # with respect to the base noise, one float64 run per coordinate: its spectral norm must stay below 20
    h = 1e-5
    J = torch.zeros(pool, D, D, dtype=torch.float64)
    lw_sens = torch.zeros(pool, dtype=torch.float64)
    for i in range(D):
        e_p = big["eps0"].double().clone()
        e_p[:, i] += h
        o64.epsilons = big["in_epsilons"].double().clone()
        o64.common_epsilon = big["in_common_epsilon"].double().clone()
        p_p, lw_p, _ = a64.sample_and_log_weights(e_p, big["noise_p"].double(), big["noise_e"].double())
        J[:, :, i] = (p_p.x - p_end.x) / h
        lw_sens = torch.maximum(lw_sens, (lw_p - lw_end).abs() / h)
    amp = torch.linalg.matrix_norm(J, ord=2)
    if verbose:
        print(f"g16: amplification of the starting state over the chain: median {float(amp.median()):.1f}, "
              f"90 % {float(amp.quantile(0.9)):.1f}, max {float(amp.max()):.0f}")
    good &= (amp <= 20.0) & (lw_sens * 3e-5 <= 5e-3)
    idx_rej = torch.nonzero(good & (rejects > 0)).flatten()
    idx_acc = torch.nonzero(good & (rejects == 0)).flatten()
    sel = torch.cat([idx_rej, idx_acc])[:keep].sort().values
    assert sel.numel() == keep, f"only {sel.numel()} well-conditioned chains in a pool of {pool}"
    noise = (big["eps0"][sel], [t[0][sel] for t in big["noise_p"]], [t[0][sel] for t in big["noise_e"]])
    out = g14_headline_arch("unused", std=std, eps_init=eps_init, seed=seed, save=False, tune=False, noise=noise)
    for j in range(M + 1):                             # chains are independent with the tuning frozen: the re-run IS the pool's run
        assert torch.equal(out["snap_x"][j], big["snap_x"][j][sel])
    rej_frac = float((out["snap_x"][1:] == out["snap_x"][:-1]).all(2).float().mean())
    if verbose:
        print(f"g16: {int(good.sum())} of {pool} chains well-conditioned throughout, {idx_rej.numel()} of them with a rejection; "
              f"kept {keep}: {rej_frac:.3f} of their proposals rejected")
    assert rej_frac >= 0.2
    out.update(pool=pool, pool_rows=sel, rejected_fraction=rej_frac)
    npz(name, **out)
    return out


def g17_step_size_trajectory(name="g17_step_size_trajectory.npz", std=0.01, seed=170, B=64, calls=6):
    """a MULTI-CALL step-size trajectory at the headline architecture (D = 32, 10 x (16-320-320-32), M = 8,
    L = 5) from the shipped `init_step_size: 1.0` (experiments/config/many_well.yaml:24-29) with tuning ON: the reference's
    `sample_and_log_weights` called `calls` times in a row on ONE sampler, so that every call starts from the step sizes the
    previous one left (hmc.py:90-100,162-170).  At this step size nearly every proposal is rejected (p_accept ~ 0): the rule
    divides on every transition, the chains stay where the flow put them, and an accept decision is never close to its threshold -
    the fixture therefore needs no selection of chains (asserted below: the smallest reject margin over all calls).  Stored per call:
    base noise, momenta, Exp(1) draws, incoming / outgoing step sizes, the returned point and log-weights."""
    from helpers import seeded_oracle_flow
    D, K, nodes, M, L, alpha = 32, 10, 10, 8, 5, 2.0
    nf = seeded_oracle_flow(D, K, nodes, seed, std)
    target = ManyWellEnergy(dim=D, use_gpu=False)
    hmc = tuned_hmc(M, D, nf, target, 1.0, L, 1, alpha, False, tune=True)
    base = EpsFlow(nf, None)
    ais = AnnealedImportanceSampler(base, target.log_prob, hmc, p_target=False, alpha=alpha, n_intermediate_distributions=M)
    torch.manual_seed(seed + 1)
    rec = {k: [] for k in ("eps0", "noise_p", "noise_e", "in_epsilons", "in_common_epsilon", "out_epsilons", "out_common_epsilon",
                           "out_x", "out_log_q", "out_log_p", "log_w", "p_accept_first", "moved")}
    margins = []
    orig_accept = hmc.metropolis_accept

    def recording_accept(point_proposed, point_current, p_proposed, p_current, beta):        # hmc.py:105-124
        with Capture() as c2:
            out = orig_accept(point_proposed, point_current, p_proposed, p_current, beta)
        recording_accept.expo.append(c2.expo[-1])
        return out
    for c in range(calls):
        base.eps = torch.randn(B, D)
        rec["eps0"].append(base.eps.clone())
        rec["in_epsilons"].append(hmc.epsilons.clone()); rec["in_common_epsilon"].append(hmc.common_epsilon.clone())
        with Capture() as cap:
            pt, log_w = ais.sample_and_log_weights(B)
        info = ais.get_logging_info()
        with torch.no_grad():
            x0, _ = nf.sample_eps(base.eps)
        rec["noise_p"].append(torch.stack(cap.randn_like)[:, None]); rec["noise_e"].append(torch.stack(cap.expo)[:, None])
        rec["out_epsilons"].append(hmc.epsilons.clone()); rec["out_common_epsilon"].append(hmc.common_epsilon.clone())
        rec["out_x"].append(pt.x.clone()); rec["out_log_q"].append(pt.log_q.clone()); rec["out_log_p"].append(pt.log_p.clone())
        rec["log_w"].append(log_w.clone()); rec["p_accept_first"].append(torch.tensor(float(info["dist0_p_accept_0"])))
        rec["moved"].append((pt.x != x0).any(1))
        assert pt.x.shape[0] == B, "a chain was dropped: pick another seed"
    out = {k: torch.stack(v) for k, v in rec.items()}
    frac_moved = float(out["moved"].float().mean())
    print(f"g17: {calls} calls from epsilon 1.0: common_epsilon {float(out['in_common_epsilon'][0]):.4f} -> "
          f"{float(out['out_common_epsilon'][-1]):.4f}, epsilons[0] {float(out['in_epsilons'][0][0, 0]):.4f} -> "
          f"{float(out['out_epsilons'][-1][0, 0]):.4f}; p_accept (first transition) per call "
          f"{[round(float(v), 5) for v in out['p_accept_first']]}; chains that moved at all: {frac_moved:.3f}")
    # every transition of every call divided its step size (the mean acceptance stayed far below 0.65, hmc.py:165-170)
    for c in range(calls):
        assert torch.allclose(out["out_epsilons"][c], out["in_epsilons"][c] / 1.05)
        assert torch.allclose(out["out_common_epsilon"][c], out["in_common_epsilon"][c] / 1.02 ** M)
        if c:
            assert torch.equal(out["in_epsilons"][c], out["out_epsilons"][c - 1])
    out.update(D=D, K=K, nodes=nodes, flow_seed=seed, flow_std=std, M=M, L=L, alpha=alpha, p_target=0, B_space=ais.B_space,
               flow_probe=torch.stack([nf.flows[0].flows[1].param_map.net[2].weight[0, :8].detach(),
                                       nf.flows[-2].flows[1].param_map.net[4].weight[1, :8].detach()]))
    npz(name, **out)
    return out


def g9_buffer():
    """Deterministic part of the reference's PrioritisedReplayBuffer (add ring wrap-around, adjust incl. the
    invalid-entry kill); sampling itself is random and is tested through properties."""
    from fab.utils.prioritised_replay_buffer import PrioritisedReplayBuffer
    torch.manual_seed(90)
    dim, L = 3, 20
    batches = [(torch.randn(8, dim), torch.randn(8), torch.randn(8)) for _ in range(4)]
    it = iter(batches)
    buf = PrioritisedReplayBuffer(dim, L, 10, lambda: next(it), fill_buffer_during_init=True)   # consumes 2 batches
    buf.add(*batches[2]); buf.add(*batches[3])                                                   # wraps around
    idx = torch.tensor([0, 5, 7, 19, 12])
    adj = torch.tensor([0.5, float("nan"), -1.0, float("inf"), 2.0])
    lq = torch.tensor([1.0, 2.0, float("nan"), 4.0, 5.0])
    buf.adjust(adj, lq, idx)
    out = dict(dim=dim, max_length=L, min_sample_length=10, idx=idx, adj=adj, lq=lq,
               x=buf.buffer.x, log_w=buf.buffer.log_w, log_q_old=buf.buffer.log_q_old,
               current_index=buf.current_index, is_full=int(buf.is_full), can_sample=int(buf.can_sample))
    for k, (x, lw, lqo) in enumerate(batches):
        out[f"b{k}_x"], out[f"b{k}_lw"], out[f"b{k}_lq"] = x, lw, lqo
    npz("g9_buffer.npz", **out)


def g10_manywell_eval():
    """ManyWellEnergy evaluation helpers (many_well.py:24-36, 96-147): the mode test set, the log-Z metrics on a
    seeded log_w, the log_q-dependent metrics with a fixed analytic log_q_fn, moments of the exact sampler."""
    target = ManyWellEnergy(dim=6, use_gpu=False)
    g = torch.Generator().manual_seed(5)
    log_w = float(target.log_Z) + 0.3 * torch.randn(5000, generator=g)
    info = target.performance_metrics(None, log_w)

    def log_q_fn(x):                                          # N(0, 1.5^2 I)
        return -0.5 * (x / 1.5).pow(2).sum(-1) - x.shape[-1] * np.log(1.5 * np.sqrt(2 * np.pi))
    torch.manual_seed(6)
    info_q = target.performance_metrics(None, log_w, log_q_fn, batch_size=2000)
    torch.manual_seed(7)
    xs = target.sample((200000,))
    npz("g10_manywell_eval.npz", modes=target._test_set_modes, log_w=log_w, log_Z=float(target.log_Z),
        relative_MSE_Z_estimate=info["relative_MSE_Z_estimate"], abs_MSE_log_Z_estimate=info["abs_MSE_log_Z_estimate"],
        test_set_modes_mean_log_prob=info_q["test_set_modes_mean_log_prob"],
        test_set_exact_mean_log_prob=info_q["test_set_exact_mean_log_prob"], forward_kl=info_q["forward_kl"],
        eval_batch_size=info_q["eval_batch_size"], sample_mean=xs.mean(0), sample_std=xs.std(0),
        sample_frac_deep_well=(xs[:, 0::2] > 0).float().mean())


def g11_gmm_eval():
    """GMM evaluation helpers (gmm.py:68-100, utils/numerical.py:25-64): quadratic_function on fixed points,
    the importance-weighted expectation bias given the constructor's true_expectation, effective_sample_size_over_p."""
    from fab.utils.numerical import quadratic_function, effective_sample_size_over_p
    torch.manual_seed(0)
    target = GMM(dim=2, n_mixes=40, loc_scaling=40.0, log_var_scaling=1.0, use_gpu=False,
                 true_expectation_estimation_n_samples=int(2e5))
    g = torch.Generator().manual_seed(3)
    x = 30 * torch.randn(4000, 2, generator=g)
    log_w = torch.randn(4000, generator=g)
    info = target.performance_metrics(x, log_w)
    npz("g11_gmm_eval.npz", locs=target.locs, x=x, log_w=log_w, fx=quadratic_function(x),
        true_expectation=target.true_expectation, bias_normed=info["bias_normed"],
        bias_no_correction=info["bias_no_correction"],
        ess_over_p=effective_sample_size_over_p(0.5 * log_w))


def g12_trainer_traces():
    """R14: the reference's PrioritisedBufferTrainer.run (fab/train_with_prioritised_buffer.py:106-255) driven for 5
    iterations on ManyWell-6 with the oracle flow as the trainable distribution, seeds {0, 1, 2}; every random draw
    (flow base noise, HMC momenta / Exp(1), the buffer's Gumbel noise and permutation) is captured together with the
    per-iteration loss, grad_norm, sampled indices and the buffer's log_w / log_q_old after the on-the-fly adjust."""
    import torch.nn as nn
    import fab.utils.prioritised_replay_buffer as prb
    from fab.utils.prioritised_replay_buffer import PrioritisedReplayBuffer
    from fab.train_with_prioritised_buffer import PrioritisedBufferTrainer
    from fab.utils.logging import ListLogger
    from fab import FABModel
    from torch.distributions.transformed_distribution import TransformedDistribution

    class OracleTrainable(nn.Module):
        """TrainableDistribution adapter (fab/trainable_distributions/base.py:4) around the oracle RealNVP."""

        def __init__(self, nf):
            super().__init__()
            self.nf = nf

        def sample_and_log_prob(self, shape):
            eps = torch.randn(shape[0], self.nf.q0.shape[0])             # captured (torch.randn)
            with torch.no_grad():
                return self.nf.sample_eps(eps)

        def sample(self, shape):
            return self.sample_and_log_prob(shape)[0]

        def log_prob(self, x):
            return self.nf.log_prob(x)

        @property
        def event_shape(self):
            return self.nf.q0.shape

    D, K, nodes, M, L, B, alpha, n_iter, n_batches = 6, 3, 5, 4, 5, 64, 2.0, 5, 2
    buf_len, buf_min = 8 * B, 2 * B
    for seed in (0, 1, 2):
        nf = make_flow(D, K, nodes, seed=60 + seed)
        flow = OracleTrainable(nf)
        target = ManyWellEnergy(dim=D, use_gpu=False)
        hmc = HamiltonianMonteCarlo(n_ais_intermediate_distributions=M, dim=D, base_log_prob=flow.log_prob,
                                    target_log_prob=target.log_prob, alpha=alpha, p_target=False, epsilon=0.2,
                                    n_outer=1, L=L)
        model = FABModel(flow=flow, target_distribution=target, n_intermediate_distributions=M, alpha=alpha,
                         transition_operator=hmc)
        out = dict(D=D, K=K, nodes=nodes, M=M, L=L, B=B, alpha=alpha, n_iter=n_iter, n_batches=n_batches,
                   buf_len=buf_len, buf_min=buf_min, lr=1e-3, max_gradient_norm=5.0, w_adjust_max_clip=10.0,
                   in_epsilons=hmc.epsilons.clone(), in_common_epsilon=hmc.common_epsilon.clone())
        out.update({k: v.detach().clone() for k, v in flow_state(nf).items()})     # the INITIAL parameters
        ais = model.annealed_importance_sampler
        calls = []
        orig_call = ais.sample_and_log_weights

        starts = []                                              # state at the START of every trainer iteration (teacher forcing)
        holder = {}

        def recording_call(batch_size, logging=True):
            if "opt" in holder:                                  # a trainer iteration begins with this call
                st = holder["opt"].state_dict()["state"]
                starts.append(dict(params={k: v.detach().clone() for k, v in nf.state_dict().items()},
                                   adam={i: (d["exp_avg"].clone(), d["exp_avg_sq"].clone(), float(d["step"]))
                                         for i, d in st.items()},
                                   eps=hmc.epsilons.clone(), ceps=hmc.common_epsilon.clone(),
                                   buf_x=buffer.buffer.x.clone(), buf_index=int(buffer.current_index),
                                   buf_full=int(buffer.is_full)))
            with Capture() as cap:
                res = orig_call(batch_size, logging)
            calls.append(dict(eps0=cap.randn[0], noise_p=torch.stack(cap.randn_like)[:, None],
                              noise_e=torch.stack(cap.expo)[:, None], x=res[0].x.clone(), log_w=res[1].clone(),
                              log_q=res[0].log_q.clone()))
            return res
        ais.sample_and_log_weights = recording_call
        gumbel, perms = [], []
        o_ts, o_rp = TransformedDistribution.sample, torch.randperm

        def ts(self_, shape=torch.Size()):
            z = o_ts(self_, shape); gumbel.append(z.clone()); return z

        def rp(*a, **k):
            p_ = o_rp(*a, **k); perms.append(p_.clone()); return p_
        torch.manual_seed(1000 + seed)

        def initial_sampler():
            pt, lw = ais.sample_and_log_weights(B, logging=False)
            return pt.x, lw, pt.log_q
        buffer = PrioritisedReplayBuffer(dim=D, max_length=buf_len, min_sample_length=buf_min,
                                         initial_sampler=initial_sampler)
        n_init_calls = len(calls)
        opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
        holder["opt"] = opt
        logger = ListLogger()
        trainer = PrioritisedBufferTrainer(model=model, optimizer=opt, buffer=buffer, alpha=alpha,
                                           n_batches_buffer_sampling=n_batches, logger=logger,
                                           max_gradient_norm=5.0, w_adjust_max_clip=10.0)
        snaps = []
        o_write = logger.write

        def write(info):
            o_write(info)
            snaps.append((buffer.buffer.log_w.clone(), buffer.buffer.log_q_old.clone()))
        logger.write = write
        o_sample = buffer.sample
        idx_log = []

        def sample(batch_size):
            r = o_sample(batch_size); idx_log.append(r[3].clone()); return r
        buffer.sample = sample
        TransformedDistribution.sample, torch.randperm = ts, rp
        try:
            trainer.run(n_iterations=n_iter, batch_size=B, save=False)
        finally:
            TransformedDistribution.sample, torch.randperm = o_ts, o_rp
        assert len(calls) == n_init_calls + n_iter and len(gumbel) == n_iter == len(perms) == len(idx_log)
        out["n_init_calls"] = n_init_calls
        for c, d in enumerate(calls):
            for k, v in d.items():
                out[f"call{c}_{k}"] = v
        hist = logger.history
        for it in range(n_iter):
            out[f"it{it}_gumbel"], out[f"it{it}_perm"], out[f"it{it}_indices"] = gumbel[it], perms[it], idx_log[it]
            out[f"it{it}_buf_log_w"], out[f"it{it}_buf_log_q_old"] = snaps[it]
            for key in ("loss", "grad_norm", "ess_ais", "ess_base", "log_Z", "w_adjust_mean", "log_q_x_mean",
                        "sampled_log_w_mean"):
                out[f"it{it}_{key}"] = hist[key][it]
        out["out_epsilons"], out["out_common_epsilon"] = hmc.epsilons, hmc.common_epsilon
        out.update({"final." + k: v for k, v in nf.state_dict().items()})
        npz(f"g12_trainer_seed{seed}.npz", **out)
        # start-of-iteration states (parameters, Adam moments, step sizes, buffer positions) in a second file: the GPU test
        # restarts every iteration from the REFERENCE's state and compares one iteration at the north-star 1e-4
        assert len(starts) == n_iter
        tf = dict(n_iter=n_iter)
        for it, st in enumerate(starts):
            tf.update({f"it{it}_param.{k}": v for k, v in st["params"].items()})
            for i, (m_, v_, step) in st["adam"].items():
                tf[f"it{it}_adam_m.{i}"], tf[f"it{it}_adam_v.{i}"], tf[f"it{it}_adam_step.{i}"] = m_, v_, step
            tf[f"it{it}_eps"], tf[f"it{it}_ceps"] = st["eps"], st["ceps"]
            tf[f"it{it}_buf_x"], tf[f"it{it}_buf_index"], tf[f"it{it}_buf_full"] = st["buf_x"], st["buf_index"], st["buf_full"]
        tf["n_adam"] = len(starts[-1]["adam"])
        npz(f"g12_trainer_seed{seed}_starts.npz", **tf)


def g13_trained_flow():
    """SURVEY 8(d) "meaningful ESS" fixture: a SMALL trained flow (ManyWell-6, RealNVP 4 x W=30, ~4k parameters) trained
    here with the reference's own PrioritisedBufferTrainer driving the oracle flow, its tuned HMC step sizes, and ONE
    evaluation AIS call of the reference (1024 chains, M = 4, p^2/q target, step sizes frozen) with all noise captured:
    ESS >> 1/B, so "ESS within 1 %" (north_star) is a real check."""
    import torch.nn as nn
    from fab.utils.prioritised_replay_buffer import PrioritisedReplayBuffer
    from fab.train_with_prioritised_buffer import PrioritisedBufferTrainer
    from fab.utils.logging import ListLogger
    from fab import FABModel

    class OracleTrainable(nn.Module):
        def __init__(self, nf):
            super().__init__()
            self.nf = nf

        def sample_and_log_prob(self, shape):
            with torch.no_grad():
                return self.nf.sample_eps(torch.randn(shape[0], self.nf.q0.shape[0]))

        def sample(self, shape):
            return self.sample_and_log_prob(shape)[0]

        def log_prob(self, x):
            return self.nf.log_prob(x)

        @property
        def event_shape(self):
            return self.nf.q0.shape

    D, K, nodes, M, L, B, alpha = 6, 4, 5, 4, 5, 128, 2.0
    torch.manual_seed(7)
    nf = oflow.make_realnvp(D, K, nodes)              # init_zeros: identity flow, like the reference's experiments
    flow = OracleTrainable(nf)
    target = ManyWellEnergy(dim=D, use_gpu=False)
    hmc = HamiltonianMonteCarlo(n_ais_intermediate_distributions=M, dim=D, base_log_prob=flow.log_prob,
                                target_log_prob=target.log_prob, alpha=alpha, p_target=False, epsilon=1.0, n_outer=1, L=L)
    model = FABModel(flow=flow, target_distribution=target, n_intermediate_distributions=M, alpha=alpha,
                     transition_operator=hmc)
    ais = model.annealed_importance_sampler

    def initial_sampler():
        pt, lw = ais.sample_and_log_weights(B, logging=False)
        return pt.x, lw, pt.log_q
    buffer = PrioritisedReplayBuffer(dim=D, max_length=B * 100, min_sample_length=B * 10, initial_sampler=initial_sampler)
    opt = torch.optim.Adam(flow.parameters(), lr=2e-3)
    trainer = PrioritisedBufferTrainer(model=model, optimizer=opt, buffer=buffer, alpha=alpha,
                                       n_batches_buffer_sampling=4, logger=ListLogger(), max_gradient_norm=100.0)
    trainer.run(n_iterations=1500, batch_size=B, save=False)
    # one evaluation call of the reference: frozen step sizes, captured noise, 1024 chains
    hmc.set_eval_mode(True)
    Be = 1024
    out = dict(D=D, K=K, nodes=nodes, M=M, L=L, alpha=alpha, B=Be, epsilons=hmc.epsilons.clone(),
               common_epsilon=hmc.common_epsilon.clone())
    out.update({k: v.detach().clone() for k, v in flow_state(nf).items()})
    for tag, p_target in (("g", False), ("p", True)):            # practical target p^2/q and target p
        model.set_ais_target(min_is_target=not p_target)
        torch.manual_seed(70 + int(p_target))
        with Capture() as cap:
            pt, lw = ais.sample_and_log_weights(Be)
        info = ais.get_logging_info()
        out.update({f"{tag}_eps0": cap.randn[0], f"{tag}_noise_p": torch.stack(cap.randn_like)[:, None],
                    f"{tag}_noise_e": torch.stack(cap.expo)[:, None], f"{tag}_log_w": lw, f"{tag}_x": pt.x,
                    f"{tag}_ess_ais": info["ess_ais"], f"{tag}_ess_base": info["ess_base"], f"{tag}_log_Z": info["log_Z"]})
        print(tag, "ess_base", info["ess_base"], "ess_ais", info["ess_ais"], "log_Z", info["log_Z"], float(target.log_Z))
    npz("g13_trained_flow_mw6.npz", **out)


GENERATORS_NOTE = "g14 needs tests/ on sys.path (helpers.seeded_oracle_flow)"
sys.path.insert(0, os.path.dirname(HERE))


if __name__ == "__main__":
    torch.set_num_threads(1)      # deterministic reduction order in the fixtures
    g1_beta(); g2_intermediate(); g3_targets(); g4_ess(); g5_multinomial()
    g6_hmc(); g7_metropolis(); g8_full_chain(); g9_buffer(); g10_manywell_eval(); g11_gmm_eval()
    g12_trainer_traces(); g13_trained_flow(); g14_headline_arch()
    g14_headline_arch("g15_ais_headline_mild.npz", std=0.01, eps_init=0.05, seed=150)
    # g16: the same architecture with BOTH accept outcomes and no ill-conditioned chain (a selection from a pool, see g16_rejecting)
    g16_rejecting()
    # g17: six calls in a row from the shipped initial step size, tuning on (the step-size state carried from call to call)
    g17_step_size_trajectory()
