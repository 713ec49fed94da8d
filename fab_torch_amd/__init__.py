"""fab_torch_amd — MI355X-native (gfx950) AIS / flow-density hot path of FAB (lollcat/fab-torch).

Python mirror of the reference's plug-in interfaces over the C ABI of libfabhip.so
(include/fabhip.h).  Importing the package does not touch the GPU; the first call into the hot path
loads (and, when hipcc is present and sources are newer, rebuilds) the HIP library and fails loudly if
that is impossible — there is no CPU fallback."""
from .point import Point
from .flow import RealNVP, make_wrapped_normflow_realnvp
from .targets import ManyWellEnergy, GMM
from .transition_operators import (TransitionOperator, HamiltonianMonteCarlo, Metropolis, create_point, grad_and_value,
                                   get_intermediate_log_prob, get_grad_intermediate_log_prob)
from .ais import AnnealedImportanceSampler, LoggingInfo, NoValidPoints
from .numerical import effective_sample_size, ess_and_log_z
from .core import FABModel
from .buffer import PrioritisedReplayBuffer, sample_without_replacement
from .train import PrioritisedBufferTrainer, Trainer
from .optim import FlatAdam
from .spline_flow import CircularCoupledRQSFlow, make_wrapped_normflow_spline
from .resample import resample, multinomial_indices, systematic_indices, multinomial_torch_compat, gather_rows



class fast_mode:
    """`fab_torch_amd.fast_mode(True)` / `with fab_torch_amd.fast_mode():` - the HMC transition / AIS kernels run the
    width x width GEMMs of the RealNVP conditioners on the bf16 matrix cores (include/fabhip.h, fabhip_set_fast_mode).
    NOT the parity path: log q differs from the fp32 kernels at the 1e-3 .. 1e-2 level.  Process-wide switch."""

    def __init__(self, on: bool = True):
        from . import _ops
        self._ops = _ops.load()
        self.prev = bool(self._ops.set_fast_mode(bool(on)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._ops.set_fast_mode(self.prev)
        return False


__all__ = [
    "Point", "RealNVP", "make_wrapped_normflow_realnvp", "ManyWellEnergy", "GMM", "TransitionOperator",
    "HamiltonianMonteCarlo", "Metropolis", "create_point", "grad_and_value", "get_intermediate_log_prob",
    "get_grad_intermediate_log_prob", "AnnealedImportanceSampler", "LoggingInfo", "NoValidPoints",
    "effective_sample_size", "ess_and_log_z", "resample", "multinomial_indices", "systematic_indices",
    "multinomial_torch_compat", "gather_rows", "FABModel", "PrioritisedReplayBuffer",
    "sample_without_replacement", "PrioritisedBufferTrainer", "Trainer", "FlatAdam", "CircularCoupledRQSFlow", "make_wrapped_normflow_spline", "fast_mode",
]
