"""Oracle ESS / log-Z / resampling — TEST INFRASTRUCTURE, never imported by the product.

* effective_sample_size ........ fab/utils/numerical.py:18-23
* log_Z ........................ fab/sampling_methods/ais.py:83-84
* resample (multinomial) ....... fab/sampling_methods/base.py:121-124 ->
  torch.distributions.Categorical.sample_n -> torch.multinomial (CPU kernel: sequential
  fp32 cumsum, divide by the total, last bucket forced to 1, float64 uniforms,
  lower-bound binary search).
* fixed-point CDF resamplers (multinomial at any N, systematic) — the build's own scalable
  definition (the reference has no systematic resampler): oracle == spec.
"""
import numpy as np
import torch
import torch.nn.functional as F

FIX_BITS = 36  # fixed-point fractional bits of the scalable CDF; sum of 2^26 weights <= 2^62


def effective_sample_size(log_w: torch.Tensor) -> torch.Tensor:
    assert log_w.dim() == 1
    w = F.softmax(log_w, dim=0)
    return 1 / torch.sum(w ** 2) / log_w.shape[0]


def log_Z(log_w: torch.Tensor, batch_size: int) -> torch.Tensor:
    lz = torch.logsumexp(log_w, dim=0)
    return lz - torch.log(torch.ones_like(lz) * batch_size)


# ---------------------------------------------------------------------------------------
# torch-compatible multinomial: exact restatement of what the reference's `resample`
# consumes from the CPU generator and returns.
# ---------------------------------------------------------------------------------------
def categorical_probs(log_w: torch.Tensor) -> torch.Tensor:
    """probs of torch.distributions.Categorical(logits=log_w) (fp32)."""
    logits = log_w - log_w.logsumexp(dim=-1, keepdim=True)
    return F.softmax(logits, dim=-1)


def multinomial_torch_compat(probs: np.ndarray, u: np.ndarray) -> np.ndarray:
    """idx_k = first j with c[j] >= u_k, c = sequential fp32 cumsum(probs)/sum, c[-1]=1."""
    probs = np.asarray(probs, dtype=np.float32)
    c = np.empty_like(probs)
    s = np.float32(0)
    for j in range(probs.shape[0]):          # sequential fp32 accumulation
        s = np.float32(s + probs[j])
        c[j] = s
    c = (c / s).astype(np.float32)
    c[-1] = np.float32(1)
    # c (fp32) compared with u (fp64) in double precision, as the CPU kernel does
    return np.searchsorted(c.astype(np.float64), np.asarray(u, np.float64), side="left").astype(np.int64)


# ---------------------------------------------------------------------------------------
# scalable fixed-point definition (associative integer prefix sums => bit-exact under any
# parallel scan order).  p_i = exp_spec(w_i - max w) — a *specified* fp32 exponential made of
# individually rounded IEEE float32 multiplies/adds (reproducible bit for bit on any machine),
# q_i = floor(p_i * 2^36) as uint64, C = inclusive prefix sum of q.
# ---------------------------------------------------------------------------------------
_EXP_C = [np.float32(c) for c in (1.5403530393381609e-4, 1.3333558146428443e-3, 9.6181291076284772e-3,
                                  5.5504108664821580e-2, 2.4022650695910071e-1, 6.9314718055994531e-1, 1.0)]


def exp_spec(x: np.ndarray) -> np.ndarray:
    """fp32: r = x*log2(e); k = rint(r); f = r-k; 2^k * Horner6(f), each op rounded to float32 (x <= 0)."""
    x = np.asarray(x, dtype=np.float32)
    r = (x * np.float32(1.44269504088896341)).astype(np.float32)
    k = np.rint(r).astype(np.float32)
    f = (r - k).astype(np.float32)
    p = np.full_like(f, _EXP_C[0])
    for c in _EXP_C[1:]:
        p = (p * f).astype(np.float32)
        p = (p + c).astype(np.float32)
    kk = np.maximum(k, np.float32(-100)).astype(np.int32)
    out = np.ldexp(p, kk).astype(np.float32)
    return np.where(k < np.float32(-60), np.float32(0), out).astype(np.float32)


def fixed_point_weights(log_w: np.ndarray) -> np.ndarray:
    lw = np.asarray(log_w, dtype=np.float32)
    finite = np.isfinite(lw)
    m = lw[finite].max() if finite.any() else np.float32(0)
    with np.errstate(invalid="ignore", over="ignore"):
        p = exp_spec(np.where(finite, lw - np.float32(m), np.float32(0)).astype(np.float32))
    p = np.where(finite, p, np.float32(0)).astype(np.float32)          # nan / +-inf -> weight 0
    return (p * np.float32(2.0 ** FIX_BITS)).astype(np.float32).astype(np.uint64)


def fixed_point_cdf(log_w: np.ndarray) -> np.ndarray:
    return np.cumsum(fixed_point_weights(log_w), dtype=np.uint64)


def _thresholds_multinomial(u: np.ndarray, total: int) -> np.ndarray:
    t = np.floor(np.asarray(u, np.float64) * np.float64(total)).astype(np.uint64)
    return np.minimum(t, np.uint64(total - 1))


def multinomial_fixed(log_w: np.ndarray, u: np.ndarray) -> np.ndarray:
    """idx_k = first j with C[j] > floor(u_k * total)."""
    C = fixed_point_cdf(log_w)
    total = int(C[-1])
    assert total > 0
    return np.searchsorted(C, _thresholds_multinomial(u, total), side="right").astype(np.int64)


def systematic_strata(total: int, n_out: int, u0: float):
    """Integer stratum map of the systematic resampler: t_k = (k * S + U) >> F with
        F = 62 - bit_length(total)            (total * 2^F in [2^61, 2^62): 64-bit arithmetic never overflows)
        S = floor(total * 2^F / n_out)         (stratum width in units of 2^-F)
        U = min(floor(u0 * float(S)), S - 1)   (the single uniform draw, u0 in [0, 1))
    i.e. n_out equal strata of width S / 2^F = total / n_out (up to a relative 2^-36 truncation, the same order as
    the 2^-36 quantisation of the weights themselves) offset by u0 strata.  Pure integer arithmetic per stratum:
    the device evaluates it - and its inverse - exactly."""
    assert total > 0 and n_out > 0 and 0.0 <= u0 < 1.0
    F = 62 - int(total).bit_length()
    S = (int(total) << F) // int(n_out)
    U = min(int(np.floor(np.float64(u0) * np.float64(S))), S - 1)
    return F, S, U


def systematic_fixed(log_w: np.ndarray, u0: float, n_out: int = None) -> np.ndarray:
    """Systematic resampling on the fixed-point CDF: idx_k = first j with C[j] > t_k, t_k = (k S + U) >> F."""
    C = fixed_point_cdf(log_w)
    total = int(C[-1])
    assert total > 0
    n_out = len(C) if n_out is None else n_out
    F, S, U = systematic_strata(total, n_out, u0)
    k = np.arange(n_out, dtype=np.uint64)
    t = (k * np.uint64(S) + np.uint64(U)) >> np.uint64(F)
    return np.searchsorted(C, t, side="right").astype(np.int64)
