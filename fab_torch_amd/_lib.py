"""ctypes binding of the C ABI of libfabhip.so (include/fabhip.h): raw device pointers + sizes, no torch type
crosses the boundary.

NOT used by the product modules - they call the TORCH_LIBRARY custom ops (`_ops.py`, csrc/torch_ops.cpp), which sit on
the same C ABI.  This file is (a) the binding a maintainer of a non-torch host would write (INTEGRATION.md quotes
it) and (b) what the test-suite uses to drive the C ABI directly: symbol/ABI checks on the CPU, and on the GPU the
test that the custom ops and the raw C ABI give bit-identical results."""
import ctypes as C
import os
import threading

import torch

from . import _build
from ._ops import FabhipError, require_device  # noqa: F401  (one exception type for both bindings)

MAX_LAYERS = 64
_FP = C.POINTER(C.c_float)


class FlowParams(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_layers", C.c_int32), ("width", C.c_int32)] + \
               [(n, C.c_void_p * MAX_LAYERS) for n in
                ("w1", "b1", "w2", "b2", "w3", "b3", "lu_L", "lu_U", "log_S", "sign_S", "perm_P")] + \
               [("loc", C.c_void_p), ("log_scale", C.c_void_p)] + \
               [("an_s", C.c_void_p * MAX_LAYERS), ("an_t", C.c_void_p * MAX_LAYERS)]


class Flow(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_layers", C.c_int32), ("width", C.c_int32), ("precision", C.c_int32),
                ("packed", C.c_void_p)]

    def __init__(self, dim=0, n_layers=0, width=0, packed=None, precision=0):      # (positional order of the old 4-field struct)
        super().__init__(dim, n_layers, width, precision, packed)


class Target(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float),
                ("log_norm", C.c_float), ("n_mix", C.c_int32), ("locs", C.c_void_p), ("scales", C.c_void_p)]


class Point(C.Structure):
    _fields_ = [("x", C.c_void_p), ("log_q", C.c_void_p), ("log_p", C.c_void_p), ("grad_log_q", C.c_void_p),
                ("grad_log_p", C.c_void_p)]


class Anneal(C.Structure):
    _fields_ = [("c_q", C.c_float), ("c_p", C.c_float), ("g_q", C.c_float), ("g_p", C.c_float)]


class HmcArgs(C.Structure):
    _fields_ = [("flow", Flow), ("target", Target), ("point", Point), ("B", C.c_int64), ("n_valid", C.c_void_p),
                ("cur", Anneal), ("next", Anneal), ("log_w", C.c_void_p), ("noise_p", C.c_void_p),
                ("noise_e", C.c_void_p), ("epsilons", C.c_void_p), ("common_epsilon", C.c_void_p),
                ("mass", C.c_void_p), ("n_outer", C.c_int32), ("L", C.c_int32), ("max_grad", C.c_float),
                ("target_p_accept", C.c_float), ("tune", C.c_int32), ("p_accept", C.c_void_p),
                ("avg_distance", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("partials", C.c_void_p)]


class MetropolisArgs(C.Structure):
    _fields_ = [("flow", Flow), ("target", Target), ("point", Point), ("B", C.c_int64), ("n_valid", C.c_void_p),
                ("cur", Anneal), ("next", Anneal), ("log_w", C.c_void_p), ("noise_x", C.c_void_p),
                ("noise_u", C.c_void_p), ("noise_scalings", C.c_void_p), ("n_updates", C.c_int32),
                ("target_p_accept", C.c_float), ("tune", C.c_int32), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class AisArgs(C.Structure):
    _fields_ = [("flow", Flow), ("target", Target), ("B", C.c_int64), ("M", C.c_int32), ("betas", C.POINTER(C.c_double)),
                ("alpha", C.c_double), ("p_target", C.c_int32), ("transition", C.c_int32), ("eps0", C.c_void_p),
                ("noise_a", C.c_void_p), ("noise_b", C.c_void_p), ("step_state", C.c_void_p),
                ("common_epsilon", C.c_void_p), ("mass", C.c_void_p), ("n_inner", C.c_int32), ("L", C.c_int32),
                ("max_grad", C.c_float), ("target_p_accept", C.c_float), ("tune", C.c_int32), ("point", Point),
                ("log_w", C.c_void_p), ("n_valid", C.c_void_p), ("stats", C.c_void_p),
                ("p_accept_first", C.c_void_p), ("p_accept_last", C.c_void_p), ("avg_distance_first", C.c_void_p),
                ("avg_distance_last", C.c_void_p), ("base_x", C.c_void_p), ("base_log_w", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


TARGET_MANYWELL, TARGET_GMM = 1, 2
TRANSITION_HMC, TRANSITION_METROPOLIS = 1, 2

_lib = None
_lock = threading.Lock()

# every symbol include/fabhip.h declares (checked by tests/test_cabi_and_host.py)
SYMBOLS = [
    "fabhip_strerror", "fabhip_version", "fabhip_flow_packed_floats", "fabhip_flow_pack", "fabhip_flow_sample",
    "fabhip_flow_log_prob", "fabhip_target_log_prob", "fabhip_create_point", "fabhip_anneal_coefs",
    "fabhip_hmc_workspace_bytes", "fabhip_hmc_transition", "fabhip_metropolis_workspace_bytes",
    "fabhip_metropolis_transition", "fabhip_ais_workspace_bytes", "fabhip_ais_run", "fabhip_ess_workspace_bytes",
    "fabhip_ess_logz", "fabhip_multinomial_torch_workspace_bytes", "fabhip_multinomial_torch",
    "fabhip_resample_workspace_bytes", "fabhip_resample_multinomial", "fabhip_resample_systematic",
    "fabhip_gather_rows", "fabhip_debug_timeline", "fabhip_flow_grad_floats", "fabhip_flow_grad_layout",
    "fabhip_flow_tape_bytes", "fabhip_flow_log_prob_tape", "fabhip_flow_param_grad", "fabhip_flow_sample_grad_tape",
    "fabhip_adam_workspace_bytes", "fabhip_adam_clip_step", "fabhip_topk_workspace_bytes", "fabhip_topk",
    "fabhip_flow_pack_density", "fabhip_abi_sizes", "fabhip_flow_tape_layout",
    "fabhip_generic_workspace_bytes", "fabhip_hmc_generic_begin", "fabhip_hmc_generic_leap_pre",
    "fabhip_hmc_generic_leap_post", "fabhip_hmc_generic_accept", "fabhip_anneal_log_prob", "fabhip_log_w_update",
    "fabhip_metropolis_generic_propose", "fabhip_metropolis_generic_accept", "fabhip_fixed_cdf",
    "fabhip_spline_packed_floats", "fabhip_spline_pack", "fabhip_spline_workspace_bytes", "fabhip_spline_log_prob",
    "fabhip_spline_sample", "fabhip_spline_tape_layout", "fabhip_spline_log_prob_tape", "fabhip_spline_sample_vjp_tape",
    "fabhip_set_fast_mode", "fabhip_get_fast_mode", "fabhip_set_option", "fabhip_get_option", "fabhip_ais_phase", "fabhip_hmc_partials_floats",
    "fabhip_hmc_adapt_gathered", "fabhip_spline_hmc_workspace_bytes", "fabhip_spline_hmc_transition",
    "fabhip_spline_ais_workspace_bytes", "fabhip_spline_ais_run", "fabhip_tape_gemm", "fabhip_debug_spline_timeline",
    "fabhip_metropolis_partials_floats", "fabhip_metropolis_adapt_gathered",
    "fabhip_flow_pack_train", "fabhip_flow_log_prob_tape_rows", "fabhip_train_step_workspace_bytes", "fabhip_buffer_train_step",
    "fabhip_buffer_add", "fabhip_buffer_sample_workspace_bytes", "fabhip_buffer_sample",
]
ABI_VERSION = 216          # FABHIP_ABI_VERSION of include/fabhip.h this binding was written against


def _declare(lib):
    i32, i64, vp, sz, dbl = C.c_int32, C.c_int64, C.c_void_p, C.c_size_t, C.c_double
    lib.fabhip_strerror.restype = C.c_char_p
    lib.fabhip_strerror.argtypes = [C.c_int]
    lib.fabhip_version.restype = C.c_int
    lib.fabhip_flow_packed_floats.restype = i64
    lib.fabhip_flow_packed_floats.argtypes = [i32, i32, i32]
    lib.fabhip_flow_pack.argtypes = [C.POINTER(FlowParams), vp, vp]
    lib.fabhip_flow_pack_density.argtypes = [C.POINTER(FlowParams), vp, vp]
    lib.fabhip_flow_sample.argtypes = [C.POINTER(Flow), vp, vp, vp, i64, vp]
    lib.fabhip_flow_log_prob.argtypes = [C.POINTER(Flow), vp, vp, vp, i64, vp]
    lib.fabhip_target_log_prob.argtypes = [C.POINTER(Target), vp, vp, vp, i64, vp]
    lib.fabhip_create_point.argtypes = [C.POINTER(Flow), C.POINTER(Target), C.POINTER(Point), i32, i64, vp]
    lib.fabhip_anneal_coefs.restype = None
    lib.fabhip_anneal_coefs.argtypes = [dbl, dbl, i32, C.POINTER(Anneal)]
    lib.fabhip_hmc_workspace_bytes.restype = sz
    lib.fabhip_hmc_workspace_bytes.argtypes = [i64, i32, i32]
    lib.fabhip_hmc_transition.argtypes = [C.POINTER(HmcArgs), vp]
    lib.fabhip_metropolis_workspace_bytes.restype = sz
    lib.fabhip_metropolis_workspace_bytes.argtypes = [i64, i32, i32]
    lib.fabhip_metropolis_transition.argtypes = [C.POINTER(MetropolisArgs), vp]
    lib.fabhip_ais_workspace_bytes.restype = sz
    lib.fabhip_ais_workspace_bytes.argtypes = [i64, i32, i32]
    lib.fabhip_ais_run.argtypes = [C.POINTER(AisArgs), vp]
    lib.fabhip_ais_phase.argtypes = [C.POINTER(AisArgs), i32, i32, i32, vp, vp]
    lib.fabhip_hmc_partials_floats.restype = i64
    lib.fabhip_hmc_partials_floats.argtypes = [i64]
    lib.fabhip_hmc_adapt_gathered.argtypes = [vp, i32, i64, vp, vp, C.c_float, i32, vp, vp, vp]
    lib.fabhip_metropolis_partials_floats.restype = i64
    lib.fabhip_metropolis_partials_floats.argtypes = [i64, i32, i32]
    lib.fabhip_metropolis_adapt_gathered.argtypes = [vp, i32, i64, i32, i32, vp, C.c_float, i32, vp]
    lib.fabhip_set_option.argtypes = [C.c_int, C.c_int]
    lib.fabhip_get_option.argtypes = [C.c_int]
    lib.fabhip_ess_workspace_bytes.restype = sz
    lib.fabhip_ess_workspace_bytes.argtypes = [i64]
    lib.fabhip_ess_logz.argtypes = [vp, i64, vp, dbl, vp, vp, sz, vp]
    lib.fabhip_multinomial_torch_workspace_bytes.restype = sz
    lib.fabhip_multinomial_torch_workspace_bytes.argtypes = [i64]
    lib.fabhip_multinomial_torch.argtypes = [vp, i64, vp, i64, vp, vp, sz, vp]
    lib.fabhip_resample_workspace_bytes.restype = sz
    lib.fabhip_resample_workspace_bytes.argtypes = [i64]
    lib.fabhip_resample_multinomial.argtypes = [vp, i64, vp, i64, vp, vp, sz, vp]
    lib.fabhip_resample_systematic.argtypes = [vp, i64, dbl, i64, vp, vp, sz, vp]
    lib.fabhip_fixed_cdf.argtypes = [vp, i64, i32, vp, vp, sz, vp]
    lib.fabhip_gather_rows.argtypes = [vp, vp, vp, i64, i64, vp]
    lib.fabhip_debug_timeline.argtypes = [vp, i32]
    lib.fabhip_flow_grad_floats.restype = i64
    lib.fabhip_flow_grad_floats.argtypes = [i32, i32, i32]
    lib.fabhip_flow_grad_layout.argtypes = [i32, i32, i32, C.POINTER(i64)]
    lib.fabhip_flow_tape_bytes.restype = sz
    lib.fabhip_flow_tape_bytes.argtypes = [i32, i32, i32, i64]
    lib.fabhip_generic_workspace_bytes.restype = sz
    lib.fabhip_generic_workspace_bytes.argtypes = [i64, i32]
    lib.fabhip_spline_packed_floats.restype = i64
    lib.fabhip_spline_packed_floats.argtypes = [i32, i32, i32]
    lib.fabhip_spline_tape_layout.argtypes = [i32, i32, i32, i64, C.POINTER(i64)]
    lib.fabhip_flow_tape_layout.argtypes = [i32, i32, i32, i64, C.POINTER(i64)]
    lib.fabhip_flow_log_prob_tape.argtypes = [C.POINTER(Flow), vp, vp, vp, i64, vp, sz, vp]
    lib.fabhip_flow_param_grad.argtypes = [C.POINTER(FlowParams), C.POINTER(Flow), vp, sz, vp, i64, vp, vp]
    lib.fabhip_flow_sample_grad_tape.argtypes = [C.POINTER(Flow), vp, vp, vp, vp, i64, vp, sz, vp]
    lib.fabhip_adam_workspace_bytes.restype = sz
    lib.fabhip_adam_workspace_bytes.argtypes = [i64]
    f32 = C.c_float
    lib.fabhip_adam_clip_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, f32, vp, vp, sz, vp]
    lib.fabhip_topk_workspace_bytes.restype = sz
    lib.fabhip_topk_workspace_bytes.argtypes = [i64, i64]
    lib.fabhip_topk.argtypes = [vp, i64, i64, i32, vp, vp, vp, sz, vp]
    for name in SYMBOLS:
        getattr(lib, name)                    # AttributeError here = header and library disagree
    # a stale binary must never be driven with newer struct layouts: ABI revision + sizeof() of every argument struct
    lib.fabhip_abi_sizes.restype = None
    lib.fabhip_abi_sizes.argtypes = [C.POINTER(i64)]
    ver = lib.fabhip_version()
    if ver != ABI_VERSION:
        raise FabhipError(f"libfabhip.so has ABI revision {ver}, this binding expects {ABI_VERSION}: rebuild it "
                          "(python -m fab_torch_amd._build --force)")
    sizes = (i64 * 8)()
    lib.fabhip_abi_sizes(sizes)
    mine = [C.sizeof(t) for t in (FlowParams, Flow, Target, Point, Anneal, HmcArgs, MetropolisArgs, AisArgs)]
    if list(sizes) != mine:
        raise FabhipError(f"struct layouts differ between libfabhip.so {list(sizes)} and the ctypes binding {mine}")
    return lib


def load():
    """Load (building first when the sources are newer and hipcc is present) libfabhip.so."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB
        if _build.is_stale():
            try:
                _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001
                # never fall back to a binary built from other sources (kernel edits / struct changes would run
                # against old code); FABHIP_ALLOW_STALE=1 is a developer escape hatch only
                if not os.path.exists(path) or os.environ.get("FABHIP_ALLOW_STALE") != "1":
                    raise FabhipError(
                        "libfabhip.so is missing or stale (sources changed since it was built) and could not be "
                        f"rebuilt — the fab_torch_amd hot path has no CPU fallback ({e})") from e
        try:
            _lib = _declare(C.CDLL(path))
        except OSError as e:
            raise FabhipError(f"cannot load {path}: {e} (no CPU fallback exists)") from e
        return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().fabhip_strerror(rc).decode()
        raise FabhipError(f"fabhip {what} failed: {msg} (code {rc})")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous float32/float64/int tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "fabhip needs contiguous tensors"
    return C.c_void_p(t.data_ptr())


class Workspace:
    """Grow-only byte workspace per (device, HIP stream) — the caller-owned scratch of the C ABI.  Keyed by the
    stream the work is enqueued on, so two streams (an evaluation stream next to the training stream, two threads)
    never share scratch; a grown buffer replaces the old one only for its own stream, whose earlier kernels are
    ordered before the next use, and the old tensor goes back to torch's caching allocator, which itself only
    reuses a block on the stream it was allocated on."""

    def __init__(self):
        self._buf = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        dev = torch.device(device)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = (str(dev), stream)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=dev)
            self._buf[key] = buf
        return buf
