"""Resampling on the GPU — `resample(x_or_point, log_w)` of fab/sampling_methods/base.py:121-124
(multinomial through torch.multinomial in the reference) plus a systematic resampler.

* `resample(..., method="multinomial")`: scalable fixed-point CDF (single-pass decoupled-look-back
  scan) + two-level binary search + vectorised row gather; uniforms are float64 draws from torch's
  device generator.
* `multinomial_torch_compat(probs, u)`: bit-exact restatement of torch's CPU multinomial given the
  probabilities and the float64 uniforms it consumed (parity with the reference's RNG path).
"""
from typing import Union

import torch

from . import _ops
from .point import Point


def multinomial_indices(log_w: torch.Tensor, n_samples: int = None, u: torch.Tensor = None) -> torch.Tensor:
    _ops.require_device(log_w, "log_w")
    lw = log_w.detach().contiguous().float()
    ns = lw.shape[0] if n_samples is None else int(n_samples)
    if u is None:
        u = torch.rand(ns, dtype=torch.float64, device=lw.device)
    return _ops.load().resample_multinomial(lw, u.contiguous().double())


def systematic_indices(log_w: torch.Tensor, u0: float = None, n_samples: int = None) -> torch.Tensor:
    _ops.require_device(log_w, "log_w")
    lw = log_w.detach().contiguous().float()
    ns = lw.shape[0] if n_samples is None else int(n_samples)
    if u0 is None:
        u0 = float(torch.rand((), dtype=torch.float64))
    return _ops.load().resample_systematic(lw, float(u0), ns)


def multinomial_torch_compat(probs: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    _ops.require_device(probs, "probs")
    return _ops.load().multinomial_torch(probs.detach().contiguous().float(), u.contiguous().double())


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _ops.require_device(src, "src")
    return _ops.load().gather_rows(src.contiguous().float(), idx.contiguous())


def resample(x_or_point: Union[Point, torch.Tensor], log_w: torch.Tensor, method: str = "multinomial"):
    """Resample points according to the log weights (same call shape as the reference)."""
    idx = multinomial_indices(log_w) if method == "multinomial" else systematic_indices(log_w)
    if isinstance(x_or_point, Point):
        p = x_or_point
        g = lambda t: None if t is None else gather_rows(t, idx)
        return Point(g(p.x), g(p.log_q), g(p.log_p), g(p.grad_log_q), g(p.grad_log_p))
    return gather_rows(x_or_point, idx)
