"""Generic plug-in path: AIS / HMC / Metropolis with ARBITRARY `Distribution` / `LogProbFunc` plug-ins
(the configuration of the reference's own tests, fab/sampling_methods/ais_test.py:86-126: a GMM target and a Gaussian
`WrappedTorchDist` base, HMC with n_outer = 5 / L = 5, Metropolis with n_updates = 5) against the CPU oracle driven
with the SAME callables and noise.  The plug-ins evaluate their own densities (torch code + autograd); the
transitions, the log-weight arithmetic and ESS / log Z are fabhip kernels."""
import pytest
import torch

from helpers import close, worst, RTOL

pytestmark = pytest.mark.gpu

fa = pytest.importorskip("fab_torch_amd")
from oracle import ais as oais            # noqa: E402

DEV = "cuda"


class TorchGMM:
    """fab/target_distributions/gmm.py:12-66 written with torch.distributions like the reference (NOT the native
    fabhip GMM): an arbitrary LogProbFunc for the generic path, usable on either device."""

    def __init__(self, dim, n_mixes, loc_scaling, device, log_var_scaling=0.1, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.locs = ((torch.rand((n_mixes, dim), generator=g) - 0.5) * 2 * loc_scaling).to(device)
        self.scale = torch.nn.functional.softplus(torch.ones(n_mixes, dim) * log_var_scaling).to(device)
        self.cat = torch.ones(n_mixes, device=device)

    def log_prob(self, x):
        mix = torch.distributions.Categorical(self.cat)
        com = torch.distributions.Independent(torch.distributions.Normal(self.locs, self.scale, validate_args=False), 1)
        lp = torch.distributions.MixtureSameFamily(mix, com, validate_args=False).log_prob(x)
        return torch.where(lp < -1e4, torch.full_like(lp, -float("inf")), lp)


class EpsGaussian:
    """N(0, 3^2 I) base with preset standard-normal noise (so that the oracle and the GPU run start from the same
    samples); otherwise a plain `Distribution` plug-in like fab.wrappers.torch.WrappedTorchDist."""

    def __init__(self, eps, scale=3.0):
        self.eps, self.scale = eps, scale
        self.dist = torch.distributions.MultivariateNormal(torch.zeros(eps.shape[1], device=eps.device),
                                                           scale_tril=scale * torch.eye(eps.shape[1], device=eps.device))

    def sample_and_log_prob(self, shape):
        x = self.scale * self.eps[:shape[0]]
        return x, self.dist.log_prob(x)

    def sample(self, shape):
        return self.sample_and_log_prob(shape)[0]

    def log_prob(self, x):
        return self.dist.log_prob(x)

    @property
    def event_shape(self):
        return (self.eps.shape[1],)


@pytest.mark.parametrize("op_kind,p_target", [("hmc", True), ("hmc", False), ("metropolis", True)])
def test_generic_ais_matches_the_oracle_with_the_same_plugins(op_kind, p_target):
    D, M, B = 2, 6, 128
    alpha = 2.0
    g = torch.Generator().manual_seed(11)
    eps0 = torch.randn(B, D, generator=g)
    hmc = op_kind == "hmc"
    n_inner = 5
    noise_a = torch.randn(M, n_inner, B, D, generator=g)
    noise_b = (torch.empty(M, n_inner, B).exponential_(generator=g) if hmc else torch.rand(M, n_inner, B, generator=g))
    res = {}
    for where in ("cpu", DEV):
        target = TorchGMM(D, 4, 8.0, where)
        base = EpsGaussian(eps0.to(where))
        if where == "cpu":
            if hmc:
                op = oais.HMC(M, D, base.log_prob, target.log_prob, alpha=alpha, p_target=p_target, epsilon=1.0,
                              n_outer=n_inner, L=5)
            else:
                op = oais.Metropolis(M, D, base.log_prob, target.log_prob, n_inner, alpha=alpha, p_target=p_target)
            ais = oais.AIS(lambda e: base.sample_and_log_prob((e.shape[0],)), base.log_prob, target.log_prob, op, p_target,
                           alpha, M, "geometric")
            pt, lw, info = ais.sample_and_log_weights(eps0, noise_a, noise_b)
            res[where] = (pt, lw, info.ess_ais, info.log_Z, op)
        else:
            if hmc:
                op = fa.HamiltonianMonteCarlo(M, D, base.log_prob, target.log_prob, alpha=alpha, p_target=p_target,
                                              epsilon=1.0, n_outer=n_inner, L=5).to(DEV)
            else:
                op = fa.Metropolis(M, D, base.log_prob, target.log_prob, n_updates=n_inner, alpha=alpha,
                                   p_target=p_target).to(DEV)
            assert not op.is_native
            ais = fa.AnnealedImportanceSampler(base, target.log_prob, op, p_target, alpha, M, "geometric")
            assert not ais.is_native
            pt, lw = ais.sample_and_log_weights(B, noise_a=noise_a.to(DEV), noise_b=noise_b.to(DEV))
            li = ais.get_logging_info()
            res[where] = (pt, lw, li["ess_ais"], li["log_Z"], op)
    (po, lwo, ess_o, lz_o, oop), (ph, lwh, ess_h, lz_h, hop) = res["cpu"], res[DEV]
    assert ph.x.shape == po.x.shape
    # low-dimensional, smooth densities: whole chains (M x 5 x 5 = 150 leapfrogs at step size ~1) agree; a chain may
    # flip at an accept threshold.  The two runs evaluate the PLUG-INS with different arithmetic (torch CPU libm vs
    # torch-ROCm device math), whose 1e-7 differences the dynamics amplify: positions within 1e-3, and the scalars
    # within what a 1e-3 displacement does to them (|grad| ~ 1): 2e-3 absolute + 1e-4 relative.
    same = (ph.x.cpu() - po.x).abs().max(1).values < 1e-3
    assert int(same.sum()) >= B - 2, f"{int((~same).sum())} chains left the oracle trajectory"
    assert close(lwh.cpu()[same], lwo[same], RTOL, atol=2e-3), worst(lwh.cpu()[same], lwo[same])
    assert close(ph.log_q.cpu()[same], po.log_q[same], RTOL, atol=2e-3)
    assert close(ph.log_p.cpu()[same], po.log_p[same], RTOL, atol=2e-3)
    if same.all():
        assert abs(ess_h - ess_o) <= 0.01 * ess_o and abs(lz_h - lz_o) <= 1e-3 * max(1.0, abs(lz_o))
        if hmc:        # identical acceptance statistics => bit-identical adapted step sizes
            assert torch.equal(hop.epsilons.cpu(), oop.epsilons) and torch.equal(hop.common_epsilon.cpu(), oop.common_epsilon)
        else:
            assert torch.equal(hop.noise_scalings.cpu(), oop.noise_scalings)


def test_wrapped_torch_dist_base_with_a_native_target_runs_the_generic_path():
    """Mixed plug-ins: a native fabhip target with a generic (torch.distributions) base - the reference's
    WrappedTorchDist.  Statistical check: AIS towards p improves the ESS over plain importance sampling."""
    D, M, B = 6, 8, 512
    torch.manual_seed(0)
    from torch_dist_plugin import WrappedTorchDist
    base = WrappedTorchDist(torch.distributions.MultivariateNormal(torch.zeros(D, device=DEV),
                                                                   scale_tril=1.5 * torch.eye(D, device=DEV)))
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, base.log_prob, target.log_prob, alpha=2.0, p_target=True, epsilon=0.3, L=5).to(DEV)
    ais = fa.AnnealedImportanceSampler(base, target.log_prob, hmc, True, None, M)
    assert not ais.is_native
    for _ in range(15):                                  # let the step sizes adapt
        pt, lw = ais.sample_and_log_weights(B)
    info = ais.get_logging_info()
    assert pt.x.shape == (B, D) and torch.isfinite(lw).all()
    assert info["ess_ais"] > 2 * info["ess_base"] and 0.3 < info["dist0_p_accept_0"] < 0.95
    assert abs(info["log_Z"] - float(target.log_Z)) < 1.0
