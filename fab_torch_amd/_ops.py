"""Loader of the PyTorch-ROCm custom-op layer `torch.ops.fabhip.*` (csrc/torch_ops.cpp: TORCH_LIBRARY(fabhip) over
the C ABI of libfabhip.so, include/fabhip.h).  Every product module reaches the HIP kernels through these ops:
they are visible to the dispatcher (CUDA/HIP key only - a CPU tensor fails there), enqueue on torch's current HIP
stream, take their scratch from the caching allocator (stream-ordered) and take part in autograd where a backward
exists (`fabhip::realnvp_logprob_tape`, registered below with torch.library.register_autograd; its backward is
`fabhip::realnvp_param_grad`).

The product path FAILS LOUDLY when the extension is missing or stale: there is no CPU / ATen fallback."""
import ctypes
import os
import threading

import torch

from . import _build

ABI_VERSION = 216          # FABHIP_ABI_VERSION of include/fabhip.h the Python side was written against

TARGET_MANYWELL, TARGET_GMM = 1, 2
TRANSITION_HMC, TRANSITION_METROPOLIS = 1, 2
PRECISION_DEFAULT, PRECISION_FP32, PRECISION_FAST = 0, 1, 2          # FABHIP_PRECISION_* (per-call fast mode)


def precision_of(flow) -> int:
    """FABHIP_PRECISION_* of a flow module: its `precision` attribute (None: the process default of `fast_mode`,
    "fp32" / "fast": this flow's calls override it)."""
    p = getattr(flow, "precision", None)
    if p is None:
        return PRECISION_DEFAULT
    if p in ("fp32", "fast"):
        return PRECISION_FP32 if p == "fp32" else PRECISION_FAST
    raise FabhipError(f"flow.precision must be None, 'fp32' or 'fast' (got {p!r})")
# developer / test switches of include/fabhip.h (fabhip_set_option)
(OPT_TILE_SHAPE, OPT_R4_STREAM, OPT_SCAN_VARIANT, OPT_SYSTEMATIC_VARIANT, OPT_SPLINE_STAGED, OPT_TIMELINE,
 OPT_SPLINE_MFMA, OPT_SPLINE_LEAP, OPT_FUSED_TAIL, OPT_ADAPT_FOLD, OPT_PGRAD, OPT_TAPE_TILES) = range(12)


class FabhipError(RuntimeError):
    pass


_ops = None
_lock = threading.Lock()


class option:
    """`with _ops.option(_ops.OPT_TILE_SHAPE, 16): ...` - set one developer switch of the library (tests, tools; A/B
    variants of one computation) and restore it on exit.  Nothing in the product path uses this."""

    def __init__(self, key: int, value: int):
        self.key, self.prev = key, int(load().set_option(int(key), int(value)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        load().set_option(self.key, self.prev)
        return False


def load():
    """torch.ops.fabhip, loading (and when sources changed and hipcc is present, rebuilding) the two libraries."""
    global _ops
    if _ops is not None:
        return _ops
    with _lock:
        if _ops is not None:
            return _ops
        if _build.is_stale():
            try:
                _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001
                # never run a binary built from other sources; FABHIP_ALLOW_STALE=1 is a developer escape hatch only
                if not (os.path.exists(_build.LIB) and os.path.exists(_build.TORCH_LIB)) \
                        or os.environ.get("FABHIP_ALLOW_STALE") != "1":
                    raise FabhipError("libfabhip.so / _fabhip_torch.so are missing or stale (sources changed since they "
                                      f"were built) and could not be rebuilt - there is no CPU fallback ({e})") from e
        try:
            ctypes.CDLL(_build.LIB, mode=ctypes.RTLD_GLOBAL)
            torch.ops.load_library(_build.TORCH_LIB)
        except OSError as e:
            raise FabhipError(f"cannot load the fabhip libraries: {e} (no CPU fallback exists)") from e
        ops = torch.ops.fabhip
        ver = ops.abi_version()
        if ver != ABI_VERSION:
            raise FabhipError(f"libfabhip.so has ABI revision {ver}, the Python side expects {ABI_VERSION}: rebuild "
                              "(python -m fab_torch_amd._build --force)")
        _register_autograd()
        _register_fakes()
        _ops = ops
        return _ops


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise FabhipError(f"{what} must live on the GPU: fab_torch_amd has no CPU path (got device {t.device})")


# ---- autograd of the training op ---------------------------------------------------------------------------------
def _tape_setup_context(ctx, inputs, output):
    theta, x, packed, params, dim, n_layers, width, want_grad_x = inputs
    log_q, grad_x, tape = output
    ctx.dims = (dim, n_layers, width)
    ctx.n_params = len(params)
    ctx.have_gx = bool(want_grad_x)
    ctx.save_for_backward(packed, tape, grad_x, *params)


def _tape_backward(ctx, g_log_q, g_grad_x, g_tape):
    """d loss / d theta = sum_b g_b d log q(x_b) / d theta as ONE flat image (fabhip_flow_param_grad: weight-gradient
    GEMMs on the matrix cores + the LU chain rule); d loss / dx = g_b * d log q / dx from the forward's reverse sweep."""
    packed, tape, grad_x = ctx.saved_tensors[:3]
    params = list(ctx.saved_tensors[3:])
    coef = g_log_q.detach().contiguous().float()
    flat = torch.ops.fabhip.realnvp_param_grad(params, packed, *ctx.dims, tape, coef) if ctx.needs_input_grad[0] else None
    gx = None
    if ctx.needs_input_grad[1]:
        if not ctx.have_gx:
            raise FabhipError("realnvp_logprob_tape: x requires grad but the forward ran without want_grad_x")
        gx = coef[:, None] * grad_x
    return flat, gx, None, [None] * ctx.n_params, None, None, None, None      # (Tensor[] slot: a list of that length)


# ---- autograd of the sampling op (reparameterised baseline losses) -------------------------------------------------
def _sample_setup_context(ctx, inputs, output):
    theta, eps, packed, params, dim, n_layers, width = inputs
    x, log_q = output
    ctx.dims = (dim, n_layers, width)
    ctx.n_params = len(params)
    ctx.save_for_backward(packed, x, *params)


def _sample_backward(ctx, g_x, g_log_q):
    """One sweep x -> eps (fabhip_flow_sample_grad_tape) writes the density path's tape with the sampling direction's
    cotangents; the parameter gradients are then the same GEMMs + LU chain rule with unit coefficients."""
    packed, x = ctx.saved_tensors[:2]
    params = list(ctx.saved_tensors[2:])
    gx = torch.zeros_like(x) if g_x is None else g_x.detach().contiguous().float()
    gl = x.new_zeros(x.shape[0]) if g_log_q is None else g_log_q.detach().contiguous().float()
    tape, g_eps = torch.ops.fabhip.realnvp_sample_grad_tape(packed, *ctx.dims, x, gx, gl)
    flat = None
    if ctx.needs_input_grad[0]:
        flat = torch.ops.fabhip.realnvp_param_grad(params, packed, *ctx.dims, tape, x.new_ones(x.shape[0]))
    return flat, (g_eps if ctx.needs_input_grad[1] else None), None, [None] * ctx.n_params, None, None, None


def _register_autograd():
    torch.library.register_autograd("fabhip::realnvp_logprob_tape", _tape_backward,
                                    setup_context=_tape_setup_context)
    torch.library.register_autograd("fabhip::realnvp_sample_tape", _sample_backward,
                                    setup_context=_sample_setup_context)


PROC_NONCE = os.urandom(8).hex()      # marks state that is only valid inside this process (registered parameter sets)


def owner_token(obj):
    """Identity of `obj` in THIS process: a copy (copy.deepcopy) or an un-pickled module carries its source's cached state in
    `__dict__` - handles of the op layer's parameter-set registry among it - and must not use it."""
    return (PROC_NONCE, id(obj))


# the host thread spins on the copy's event while it waits for the end of a call (a core at 100 % for the call's few milliseconds:
# what a latency-bound sampler wants); FABHIP_POLL_READS=0 (or `_ops.POLL_READS = False`) blocks in the driver instead
POLL_READS = os.environ.get("FABHIP_POLL_READS", "1") != "0"
_pinned = threading.local()          # (per thread: the buffer is handed back to the caller)


def _read_small(t: torch.Tensor, between=None) -> torch.Tensor:
    """Device -> host copy of a few words at the END of a call the host is waiting for: into a pinned buffer, then the copy's
    event is polled (hipEventQuery) instead of sleeping in a blocking synchronise - the wake-up of a blocked thread costs tens of
    microseconds during which the GPU has nothing queued.  Polling is bounded (a call of this library lasts milliseconds); past
    the bound the thread blocks."""
    key = (t.device, t.dtype, t.numel())
    cache = _pinned.__dict__.setdefault("cache", {})
    ent = cache.get(key)
    if ent is None:
        ent = (torch.empty(t.numel(), dtype=t.dtype, pin_memory=True), torch.cuda.Event())
        cache[key] = ent
    buf, ev = ent
    # copy and event on the current stream of t's DEVICE (the ops run under a device guard, so t may live on a device that
    # is not the current one: an event recorded on the current device's stream would complete before the copy lands)
    if t.device.index == torch.cuda.current_device():   # (the common case: no guard, no stream lookup on the call's host path)
        buf.copy_(t, non_blocking=True)
        ev.record()
    else:
        with torch.cuda.device(t.device):
            buf.copy_(t, non_blocking=True)
            ev.record(torch.cuda.current_stream(t.device))
    if between is not None:                             # work to enqueue BEHIND the copy while the host waits for it
        between()
    if POLL_READS:
        for _ in range(200000):
            if ev.query():
                break
        else:
            ev.synchronize()
    else:
        ev.synchronize()
    return buf                                          # (valid until the next read of the same size on this device)


def read_counts_and_stats(n_valid: torch.Tensor, stats: torch.Tensor, between=None):
    """(stats[:6] on the host, (n_valid[0], n_valid[1])) with ONE device->host copy: the AIS ops allocate `stats` (float[16]) and
    `n_valid` (int32[2]) as views of one 18-word buffer, which is read whole; tensors from elsewhere (the phase ops of the
    sharded sampler own theirs) take the two-kernel route."""
    if stats.is_cuda and stats.dtype == torch.float32 and n_valid.dtype == torch.int32:
        st = stats.untyped_storage()
        if (n_valid.untyped_storage().data_ptr() == st.data_ptr() and st.nbytes() == 72 and stats.storage_offset() == 0
                and n_valid.storage_offset() == 16 and stats.numel() == 16 and n_valid.numel() == 2):
            h = _read_small(stats.as_strided((18,), (1,)), between)
            n = h[16:18].view(torch.int32)
            return h[:6], (int(n[0]), int(n[1]))
    if between is not None:
        between()
    h = torch.cat([n_valid.float(), stats[:6]]).cpu()
    return h[2:], (int(h[0]), int(h[1]))


# ---- shape functions (torch.compile / fake tensors) for the tensor-in / tensor-out density ops ---------------------
def _register_fakes():
    rf = torch.library.register_fake

    @rf("fabhip::realnvp_sample")
    def _(packed, dim, n_layers, width, eps):
        return torch.empty_like(eps), eps.new_empty((eps.shape[0],))

    @rf("fabhip::realnvp_logprob_grad")
    def _(packed, dim, n_layers, width, x, with_grad):
        return x.new_empty((x.shape[0],)), (torch.empty_like(x) if with_grad else x.new_empty((0,)))

    @rf("fabhip::target_logp_grad")
    def _(target_kind, target_params, locs, scales, x, with_grad):
        return x.new_empty((x.shape[0],)), (torch.empty_like(x) if with_grad else x.new_empty((0,)))

    @rf("fabhip::ess_logz")
    def _(log_w, n_ptr, n_norm):
        return log_w.new_empty((3,))

    @rf("fabhip::gather_rows")
    def _(src, idx):
        return src.new_empty((idx.shape[0],) + tuple(src.shape[1:]))

    @rf("fabhip::resample_systematic")
    def _(log_w, u0, n_samples):
        return log_w.new_empty((n_samples,), dtype=torch.int64)

    @rf("fabhip::resample_multinomial")
    def _(log_w, u):
        return log_w.new_empty((u.shape[0],), dtype=torch.int64)
