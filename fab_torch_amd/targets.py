"""Target distributions of the hot path: ManyWell (fab/target_distributions/many_well.py:16-90,
double_well.py:31-58) and the 40-mode GMM (fab/target_distributions/gmm.py:12-66) — same constructor
arguments and `log_prob` semantics; `log_prob` runs csrc/target_device.h on the GPU."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ops
from .numerical import (quadratic_function, importance_weighted_expectation,
                        effective_sample_size_over_p)


class _NativeTarget(nn.Module):
    def native_target(self):
        """The target arguments of the ops: (kind, [a, b, c, log_norm], locs or None, scales or None)."""
        raise NotImplementedError

    def _native_log_prob(self, x, with_grad=False):
        _ops.require_device(x, "x")
        lp, g = _ops.load().target_logp_grad(*self.native_target(), x.detach().contiguous().float(), bool(with_grad))
        return lp, (g if with_grad else None)

    def log_prob_and_grad(self, x):
        return self._native_log_prob(x, with_grad=True)


class _TargetLogProb(torch.autograd.Function):
    """log p(x) with d log p / dx from the same HIP launch (fabhip_target_log_prob): what generic autograd callers
    such as `grad_and_value(x, target.log_prob)` (fab/sampling_methods/base.py:50-56) and the reverse-KL baseline
    losses differentiate through."""

    @staticmethod
    def forward(ctx, target, x):
        lp, g = target._native_log_prob(x, with_grad=True)
        ctx.save_for_backward(g)
        return lp

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return None, grad_out[:, None] * g


def _target_log_prob(target, x):
    _ops.require_device(x, "x")
    if torch.is_grad_enabled() and x.requires_grad:
        return _TargetLogProb.apply(target, x)
    return target._native_log_prob(x)[0]


class ManyWellEnergy(_NativeTarget):
    """log p(x) = sum_i -(a x_{2i} + b x_{2i}^2 + c x_{2i}^4 + x_{2i+1}^2 / 2)."""

    def __init__(self, dim=4, use_gpu: bool = True, normalised: bool = False, a=-0.5, b=-6.0, c=1.0):
        super().__init__()
        assert dim % 2 == 0
        self.dim, self.n_wells = dim, dim // 2
        self._a, self._b, self._c = a, b, c
        self.normalised = normalised
        self.centre = 1.7
        self.register_buffer("_anchor", torch.zeros(1))
        if use_gpu and torch.cuda.is_available():
            self.cuda()
        self.device = "cuda" if (use_gpu and torch.cuda.is_available()) else "cpu"

    @property
    def log_Z_2D(self):
        if self._a == -0.5 and self._b == -6 and self._c == 1.0:
            return np.log(11784.50927) + 0.5 * np.log(2 * math.pi)          # double_well.py:97-101
        raise NotImplementedError

    @property
    def log_Z(self):
        return torch.tensor(self.log_Z_2D * self.n_wells)

    @property
    def Z(self):
        return torch.exp(self.log_Z)

    def native_target(self):
        log_norm = float(self.log_Z_2D * self.n_wells) if self.normalised else 0.0
        return _ops.TARGET_MANYWELL, [float(self._a), float(self._b), float(self._c), log_norm], None, None

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return _target_log_prob(self, x)

    # ---- evaluation helpers (many_well.py:61-147, double_well.py:61-95; host-side glue, torch on the device) ----
    max_dim_for_all_modes = 40

    def _eval_device(self):
        return self._anchor.device

    def sample(self, shape) -> torch.Tensor:
        """Exact samples: per well, rejection sampling of the quartic coordinate (proposal 0.2 N(-1.7, .5) +
        0.8 N(1.7, .5), envelope 3 Z as in double_well.py:61-82) and a standard normal for the other one."""
        if not (self._a == -0.5 and self._b == -6 and self._c == 1.0):
            raise NotImplementedError
        assert len(shape) == 1
        n, dev = int(shape[0]), self._eval_device()
        need = n * self.n_wells
        mix = torch.tensor([0.2, 0.8], device=dev)
        means = torch.tensor([-1.7, 1.7], device=dev)
        log_k = math.log(11784.50927 * 3)
        got = []
        have = 0
        while have < need:
            m = int((need - have) * 3.5) + 64                      # acceptance rate is 1/3
            comp = torch.multinomial(mix, m, replacement=True)
            z = means[comp] + 0.5 * torch.randn(m, device=dev)
            log_prop = torch.logsumexp(torch.log(mix)[None, :] - 0.5 * ((z[:, None] - means[None, :]) / 0.5) ** 2
                                       - math.log(0.5 * math.sqrt(2 * math.pi)), dim=1)
            log_target = -z ** 4 + 6 * z ** 2 + 0.5 * z
            keep = torch.rand(m, device=dev).log() < log_target - log_prop - log_k
            got.append(z[keep])
            have += int(keep.sum())
        x1 = torch.cat(got)[:need].view(n, self.n_wells)
        x = torch.empty(n, self.dim, device=dev)
        x[:, 0::2] = x1
        x[:, 1::2] = torch.randn(n, self.n_wells, device=dev)
        return x

    def _well(self):
        w = self.__dict__.get("_well2d")
        if w is None:
            w = self.__dict__["_well2d"] = ManyWellEnergy(2, a=self._a, b=self._b, c=self._c)
        return w

    def log_prob_2D(self, x):
        """many_well.py:92-94: the double-well density of one coordinate pair (plotting helper), HIP like log_prob."""
        return self._well().log_prob(x)

    def energy(self, x, temperature=None):
        """double_well.py:19-23 (2-D, shape [..., 1])."""
        assert x.shape[-1] == 2, "`x` does not match `dim`"
        return -self._well().log_prob(x)[..., None] / (1.0 if temperature is None else temperature)

    def force(self, x, temperature=None):
        """double_well.py:25-28: -d energy / dx."""
        assert x.shape[-1] == 2, "`x` does not match `dim`"
        _, g = self._well().log_prob_and_grad(x.detach().contiguous().float())
        return g / (1.0 if temperature is None else temperature)

    def sample_first_dimension(self, shape):
        """double_well.py:60-82: exact samples of the quartic coordinate of one well (rejection sampling)."""
        assert len(shape) == 1
        return self._well().sample(shape)[:, 0]

    def get_modes_test_set_iterator(self, batch_size: int):
        """Points placed at the modes (x_even = +-1.7, x_odd = 0): all 2^(dim/2) of them below 40 dims, 10^4 random
        ones above (many_well.py:24-36, 68-79)."""
        dev = self._eval_device()
        if self.dim < self.max_dim_for_all_modes:
            k = torch.arange(2 ** self.n_wells, device=dev)
            bits = (k[:, None] >> torch.arange(self.n_wells - 1, -1, -1, device=dev)[None, :]) & 1
        else:
            bits = torch.randint(high=2, size=(int(1e4), self.n_wells), device=dev)
        test_set = torch.zeros(bits.shape[0], self.dim, device=dev)
        test_set[:, 0::2] = -self.centre + 2 * self.centre * bits.float()
        return list(torch.split(test_set, batch_size))

    def performance_metrics(self, samples, log_w: torch.Tensor, log_q_fn=None, batch_size=None):
        """many_well.py:96-147: accuracy of the normalisation-constant estimate over 50 splits of log_w and, given
        log_q_fn, mean flow log-prob on the mode test set / on exact samples and the forward KL estimate."""
        del samples
        n_runs = 50
        n_vals = log_w.shape[0] // n_runs
        lw = torch.stack(log_w[:n_vals * n_runs].split(n_runs), dim=-1)
        log_Z = float(self.log_Z)
        log_Z_estimate = torch.logsumexp(lw, dim=-1) - np.log(lw.shape[-1])
        info = {"relative_MSE_Z_estimate": torch.mean(torch.abs(torch.exp(log_Z_estimate - log_Z) - 1)).item(),
                "abs_MSE_log_Z_estimate": torch.mean(torch.abs(log_Z_estimate - log_Z)).item()}
        if log_q_fn is not None:
            assert batch_size is not None
            n_batches = max(lw.shape[0] // batch_size, 1)
            modes = self.get_modes_test_set_iterator(batch_size)
            with torch.no_grad():
                sum_modes = sum(float(torch.sum(log_q_fn(x))) for x in modes)
                sum_exact = sum_kl = 0.0
                for _ in range(n_batches):
                    x = self.sample((batch_size,))
                    lq = log_q_fn(x)
                    sum_exact += float(torch.sum(lq))
                    sum_kl += float(torch.sum(self.log_prob(x) - log_Z - lq))       # as written in many_well.py:137
            n_eval = batch_size * n_batches
            info.update(test_set_modes_mean_log_prob=sum_modes / sum(len(x) for x in modes),
                        test_set_exact_mean_log_prob=sum_exact / n_eval, forward_kl=sum_kl / n_eval,
                        eval_batch_size=n_eval)
        return info


class GMM(_NativeTarget):
    def __init__(self, dim, n_mixes, loc_scaling, log_var_scaling=0.1, seed=0, n_test_set_samples=1000,
                 use_gpu=True, true_expectation_estimation_n_samples=int(1e7)):
        super().__init__()
        self.seed, self.n_mixes, self.dim = seed, n_mixes, dim
        self.n_test_set_samples = n_test_set_samples
        self._true_expectation, self._true_expectation_n = None, int(true_expectation_estimation_n_samples)
        mean = (torch.rand((n_mixes, dim)) - 0.5) * 2 * loc_scaling          # gmm.py:22 (caller seeds torch)
        log_var = torch.ones((n_mixes, dim)) * log_var_scaling
        self.register_buffer("cat_probs", torch.ones(n_mixes))
        self.register_buffer("locs", mean)
        self.register_buffer("scale_trils", torch.diag_embed(F.softplus(log_var)))
        self.register_buffer("scales", F.softplus(log_var).contiguous())
        self.device = "cuda" if (use_gpu and torch.cuda.is_available()) else "cpu"
        if self.device == "cuda":
            self.cuda()

    def native_target(self):
        return _ops.TARGET_GMM, [0.0, 0.0, 0.0, 0.0], self.locs, self.scales

    @property
    def distribution(self):
        mix = torch.distributions.Categorical(self.cat_probs)
        com = torch.distributions.MultivariateNormal(self.locs, scale_tril=self.scale_trils, validate_args=False)
        return torch.distributions.MixtureSameFamily(mix, com, validate_args=False)

    def sample(self, shape=(1,)):
        return self.distribution.sample(shape)

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return _target_log_prob(self, x)

    # ---- evaluation helpers (gmm.py:33-36, 54-100; host-side torch, not on the hot path) ---------------------------
    expectation_function = staticmethod(quadratic_function)

    @property
    def true_expectation(self) -> torch.Tensor:
        """E_p[quadratic_function] by plain Monte Carlo; the reference estimates it in the constructor
        (gmm.py:30-33, 10^7 samples), here on first use."""
        if self._true_expectation is None:
            n, acc, done = self._true_expectation_n, 0.0, 0
            while done < n:
                m = min(n - done, 1 << 20)
                acc += float(torch.sum(self.expectation_function(self.sample((m,))).double()))
                done += m
            self._true_expectation = torch.tensor(acc / n, dtype=torch.float32)
        return self._true_expectation

    @property
    def test_set(self) -> torch.Tensor:
        return self.sample((self.n_test_set_samples,))

    def evaluate_expectation(self, samples, log_w):
        expectation = importance_weighted_expectation(self.expectation_function, samples, log_w)
        true_expectation = self.true_expectation.to(expectation.device)
        return (expectation - true_expectation) / true_expectation

    def performance_metrics(self, samples, log_w, log_q_fn=None, batch_size=None):
        bias_normed = self.evaluate_expectation(samples, log_w)
        bias_no_correction = self.evaluate_expectation(samples, torch.ones_like(log_w))
        if log_q_fn:
            with torch.no_grad():
                log_q_test = log_q_fn(self.test_set)          # two independent draws of the test set, like the
                log_p_test = self.log_prob(self.test_set)      # reference's property (gmm.py:54-56, 88-89)
            return {"test_set_mean_log_prob": torch.mean(log_q_test).item(),
                    "bias_normed": torch.abs(bias_normed).item(),
                    "bias_no_correction": torch.abs(bias_no_correction).item(),
                    "ess_over_p": effective_sample_size_over_p(log_p_test - log_q_test).item(),
                    "kl_forward": torch.mean(log_p_test - log_q_test).item()}
        return {"bias_normed": bias_normed.item(), "bias_no_correction": torch.abs(bias_no_correction).item()}
