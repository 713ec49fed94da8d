"""Resampling on the GPU — `resample(x_or_point, log_w)` of fab/sampling_methods/base.py:121-124
(multinomial through torch.multinomial in the reference) plus a systematic resampler.

* `resample(..., method="multinomial")`: scalable fixed-point CDF (single-pass decoupled-look-back
  scan) + two-level binary search + vectorised row gather; uniforms are float64 draws from torch's
  device generator.
* `multinomial_torch_compat(probs, u)`: bit-exact restatement of torch's CPU multinomial given the
  probabilities and the float64 uniforms it consumed (parity with the reference's RNG path).
"""
import ctypes as C
from typing import Union

import torch

from . import _lib
from .point import Point

_ws = _lib.Workspace()


def _aligned_ws(nbytes, device):
    buf = _ws.get(nbytes + 256, device)
    off = (-buf.data_ptr()) % 256
    return buf.data_ptr() + off


def multinomial_indices(log_w: torch.Tensor, n_samples: int = None, u: torch.Tensor = None) -> torch.Tensor:
    lib = _lib.load()
    _lib.require_device(log_w, "log_w")
    lw = log_w.detach().contiguous().float()
    n = lw.shape[0]
    ns = n if n_samples is None else int(n_samples)
    if u is None:
        u = torch.rand(ns, dtype=torch.float64, device=lw.device)
    u = u.contiguous().double()
    idx = torch.empty(ns, dtype=torch.int64, device=lw.device)
    nb = lib.fabhip_resample_workspace_bytes(n)
    _lib.check(lib.fabhip_resample_multinomial(_lib.ptr(lw), n, _lib.ptr(u), ns, _lib.ptr(idx),
                                               C.c_void_p(_aligned_ws(nb, lw.device)), nb, _lib.stream_ptr()),
               "resample_multinomial")
    return idx


def systematic_indices(log_w: torch.Tensor, u0: float = None, n_samples: int = None) -> torch.Tensor:
    lib = _lib.load()
    _lib.require_device(log_w, "log_w")
    lw = log_w.detach().contiguous().float()
    n = lw.shape[0]
    ns = n if n_samples is None else int(n_samples)
    if u0 is None:
        u0 = float(torch.rand((), dtype=torch.float64))
    idx = torch.empty(ns, dtype=torch.int64, device=lw.device)
    nb = lib.fabhip_resample_workspace_bytes(n)
    _lib.check(lib.fabhip_resample_systematic(_lib.ptr(lw), n, float(u0), ns, _lib.ptr(idx),
                                              C.c_void_p(_aligned_ws(nb, lw.device)), nb, _lib.stream_ptr()),
               "resample_systematic")
    return idx


def multinomial_torch_compat(probs: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _lib.require_device(probs, "probs")
    p = probs.detach().contiguous().float()
    u = u.contiguous().double()
    n, ns = p.shape[0], u.shape[0]
    idx = torch.empty(ns, dtype=torch.int64, device=p.device)
    nb = lib.fabhip_multinomial_torch_workspace_bytes(n)
    ws = _ws.get(nb, p.device)
    _lib.check(lib.fabhip_multinomial_torch(_lib.ptr(p), n, _lib.ptr(u), ns, _lib.ptr(idx), _lib.ptr(ws), nb,
                                            _lib.stream_ptr()), "multinomial_torch")
    return idx


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    s = src.contiguous().float()
    s2 = s.reshape(s.shape[0], -1)
    out = torch.empty((idx.shape[0], s2.shape[1]), dtype=torch.float32, device=s.device)
    _lib.check(lib.fabhip_gather_rows(_lib.ptr(s2), _lib.ptr(idx.contiguous()), _lib.ptr(out), idx.shape[0],
                                      s2.shape[1], _lib.stream_ptr()), "gather_rows")
    return out.reshape((idx.shape[0],) + tuple(s.shape[1:]))


def resample(x_or_point: Union[Point, torch.Tensor], log_w: torch.Tensor, method: str = "multinomial"):
    """Resample points according to the log weights (same call shape as the reference)."""
    idx = multinomial_indices(log_w) if method == "multinomial" else systematic_indices(log_w)
    if isinstance(x_or_point, Point):
        p = x_or_point
        g = lambda t: None if t is None else gather_rows(t, idx)
        return Point(g(p.x), g(p.log_q), g(p.log_p), g(p.grad_log_q), g(p.grad_log_p))
    return gather_rows(x_or_point, idx)
