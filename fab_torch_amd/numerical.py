"""ESS / log-Z on the GPU — fab/utils/numerical.py:18-23 and fab/sampling_methods/ais.py:80-86."""
import torch
import torch.nn.functional as F

from . import _ops


def ess_and_log_z(log_w: torch.Tensor, n_norm: float = None) -> torch.Tensor:
    """Device float32[3]: (normalised ESS, logsumexp(log_w) - log(n_norm), n)."""
    assert log_w.dim() == 1
    _ops.require_device(log_w, "log_w")
    lw = log_w.detach().contiguous().float()
    return _ops.load().ess_logz(lw, None, float(lw.shape[0] if n_norm is None else n_norm))


def effective_sample_size(log_w: torch.Tensor, normalised=False) -> torch.Tensor:
    """Same signature as the reference; `normalised=True` means `log_w` already holds normalised weights."""
    assert len(log_w.shape) == 1
    if normalised:
        return 1 / torch.sum(log_w ** 2) / log_w.shape[0]
    return ess_and_log_z(log_w)[0]


# ---- evaluation helpers of fab/utils/numerical.py:8-15, 25-64 (host-side torch; not on the hot path) -------------
def MC_estimate_true_expectation(distribution, expectation_function, n_samples: int) -> torch.Tensor:
    return torch.mean(expectation_function(distribution.sample((n_samples,))))


def effective_sample_size_over_p(log_w: torch.Tensor) -> torch.Tensor:
    """ESS estimated with samples from the (normalised) target: 1 / mean(exp(log p - log q))."""
    assert len(log_w.shape) == 1
    return 1 / torch.mean(torch.exp(log_w))


def setup_quadratic_function(x: torch.Tensor, seed: int = 0):
    """The reference seeds the global CPU generator, draws (x_shift, A, b) and re-randomises it (numerical.py:35-47).
    Same values from a private generator; the global RNG is left alone."""
    g = torch.Generator().manual_seed(seed)
    n = x.shape[-1]
    x_shift = 2 * torch.randn(n, generator=g)
    A = 2 * torch.rand((n, n), generator=g)
    b = torch.rand(n, generator=g)
    return tuple(t.to(device=x.device, dtype=x.dtype) for t in (x_shift, A, b))


def quadratic_function(x: torch.Tensor, seed: int = 0) -> torch.Tensor:
    x_shift, A, b = setup_quadratic_function(x, seed)
    x = x + x_shift
    return torch.einsum("bi,ij,bj->b", x, A, x) + torch.einsum("i,bi->b", b, x)


def importance_weighted_expectation(f, x: torch.Tensor, log_w: torch.Tensor) -> torch.Tensor:
    return F.softmax(log_w, dim=-1) @ f(x)            # log_w is 1-D: the reference's `.T` is a no-op
