"""ESS / log-Z on the GPU — fab/utils/numerical.py:18-23 and fab/sampling_methods/ais.py:80-86."""
import torch
import torch.nn.functional as F

from . import _lib

_ws = _lib.Workspace()


def ess_and_log_z(log_w: torch.Tensor, n_norm: float = None) -> torch.Tensor:
    """Device float32[3]: (normalised ESS, logsumexp(log_w) - log(n_norm), n)."""
    lib = _lib.load()
    assert log_w.dim() == 1
    _lib.require_device(log_w, "log_w")
    lw = log_w.detach().contiguous().float()
    n = lw.shape[0]
    out = torch.empty(3, dtype=torch.float32, device=lw.device)
    nb = lib.fabhip_ess_workspace_bytes(n)
    ws = _ws.get(nb, lw.device)
    _lib.check(lib.fabhip_ess_logz(_lib.ptr(lw), n, None, float(n if n_norm is None else n_norm), _lib.ptr(out),
                                   _lib.ptr(ws), nb, _lib.stream_ptr()), "ess_logz")
    return out


def effective_sample_size(log_w: torch.Tensor, normalised=False) -> torch.Tensor:
    """Same signature as the reference; `normalised=True` means `log_w` already holds normalised weights."""
    assert len(log_w.shape) == 1
    if normalised:
        return 1 / torch.sum(log_w ** 2) / log_w.shape[0]
    return ess_and_log_z(log_w)[0]
