"""Oracle target log-densities — TEST INFRASTRUCTURE, never imported by the product.

ManyWell: fab/target_distributions/many_well.py:81-90 -> double_well.py:44-58.
GMM:      fab/target_distributions/gmm.py:22-27,57-66.
Pinned against the imported reference by tests/golden/make_golden.py (fixture G3).
"""
import math

import torch
import torch.nn.functional as F


class ManyWell:
    """log p(x) = sum_i [ -a x_{2i} - b x_{2i}^2 - c x_{2i}^4 - 0.5 x_{2i+1}^2 ]  (a=-0.5,b=-6,c=1)."""

    def __init__(self, dim: int, a: float = -0.5, b: float = -6.0, c: float = 1.0,
                 normalised: bool = False):
        assert dim % 2 == 0
        self.dim, self.a, self.b, self.c = dim, a, b, c
        self.n_wells = dim // 2
        self.normalised = normalised

    @property
    def log_Z(self):
        # double_well.py:97-101 (only for the default a, b, c)
        return (math.log(11784.50927) + 0.5 * math.log(2 * math.pi)) * self.n_wells

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        # Same op order as the reference: per well e1 + e2, negate, stack, sum over wells.
        per_well = []
        for i in range(self.n_wells):
            x1 = x[:, 2 * i]
            x2 = x[:, 2 * i + 1]
            e1 = self.a * x1 + self.b * x1.pow(2) + self.c * x1.pow(4)
            e2 = 0.5 * x2.pow(2)
            per_well.append(-(e1 + e2))
        log_prob = torch.sum(torch.stack(per_well), dim=0)
        if self.normalised:
            return log_prob - self.log_Z
        return log_prob

    def grad_log_prob(self, x: torch.Tensor) -> torch.Tensor:
        g = torch.empty_like(x)
        x1 = x[:, 0::2]
        g[:, 0::2] = -(self.a + 2 * self.b * x1 + 4 * self.c * x1.pow(3))
        g[:, 1::2] = -x[:, 1::2]
        return g


class GMM:
    """Equal-weight diagonal GMM with the reference's -inf mask below -1e4 (gmm.py:57-66)."""

    def __init__(self, dim: int, n_mixes: int, loc_scaling: float, log_var_scaling: float = 0.1,
                 seed: int = 0):
        # experiments/gmm/run.py:53 seeds torch before constructing the target.
        torch.manual_seed(seed)
        self.dim, self.n_mixes = dim, n_mixes
        self.locs = (torch.rand((n_mixes, dim)) - 0.5) * 2 * loc_scaling            # gmm.py:22
        log_var = torch.ones((n_mixes, dim)) * log_var_scaling                       # gmm.py:23
        self.scales = F.softplus(log_var)                                           # diag of scale_tril, gmm.py:27

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        # MixtureSameFamily.log_prob = logsumexp_k( log_softmax(cat_logits)_k + N_k(x) )
        diff = (x[:, None, :] - self.locs[None]) / self.scales[None]                 # [B, K, D]
        half_log_det = self.scales.log().sum(-1)                                     # [K]
        comp = -0.5 * (self.dim * math.log(2 * math.pi) + diff.pow(2).sum(-1)) - half_log_det
        log_mix = -math.log(self.n_mixes)
        lp = torch.logsumexp(comp + log_mix, dim=-1)
        mask = torch.zeros_like(lp)
        mask[lp < -1e4] = -float("inf")
        return lp + mask
