"""AnnealedImportanceSampler with the reference's interface (fab/sampling_methods/ais.py:20-213).

`sample_and_log_weights(batch_size)` is ONE C-ABI call (fabhip_ais_run): flow sample, point creation,
initial log-weights, NaN/inf compaction, M fused transitions with log-weight accumulation, final
compaction and ESS / log Z — enqueued back-to-back on the current HIP stream with a single small
device->host read at the end (row counts + logging scalars)."""
import ctypes as C
from typing import Any, Dict, NamedTuple, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .flow import RealNVP
from .point import Point
from .targets import _NativeTarget
from .transition_operators import HamiltonianMonteCarlo, Metropolis, TransitionOperator, create_point, _owner


class LoggingInfo(NamedTuple):
    ess_base: float
    ess_ais: float
    log_Z: float


class AnnealedImportanceSampler:
    def __init__(self, base_distribution, target_log_prob, transition_operator: TransitionOperator,
                 p_target: bool, alpha: Optional[float] = None, n_intermediate_distributions: int = 1,
                 distribution_spacing_type: str = "linear"):
        if not p_target:
            assert alpha is not None, "Must specify alpha if AIS target is not p."
        self.base_distribution = base_distribution
        self.target_log_prob = target_log_prob
        self.transition_operator = transition_operator
        self.p_target = p_target
        self.alpha = alpha
        self.n_intermediate_distributions = n_intermediate_distributions
        self.distribution_spacing_type = distribution_spacing_type
        self.B_space = self.setup_distribution_spacing(distribution_spacing_type, n_intermediate_distributions)
        self._logging_info: LoggingInfo
        self._ws = _lib.Workspace()
        self._last_stats = None

    def get_logging_info(self) -> Dict[str, Any]:
        info = self._logging_info._asdict()
        info.update(self.transition_operator.get_logging_info())
        return info

    def setup_distribution_spacing(self, distribution_spacing_type: str, n_intermediate_distributions: int
                                   ) -> torch.Tensor:
        assert n_intermediate_distributions > 0
        if distribution_spacing_type == "geometric":
            n_lin = int(n_intermediate_distributions / 4)
            n_geo = n_intermediate_distributions - n_lin - 1
            B_space = np.concatenate([np.linspace(0, 0.01, n_lin + 2)[:-1], np.geomspace(0.01, 1, n_geo + 2)])
        elif distribution_spacing_type == "linear":
            B_space = np.linspace(0.0, 1.0, n_intermediate_distributions + 2)
        else:
            raise Exception(f"distribution spacing incorrectly specified: '{distribution_spacing_type}',"
                            f"options are 'geometric' or 'linear'")
        assert B_space.shape == (self.n_intermediate_distributions + 2,)
        return torch.tensor(B_space)

    # ---------------------------------------------------------------------------------------------
    def _native_parts(self) -> Tuple[RealNVP, _NativeTarget]:
        flow = self.base_distribution
        if not isinstance(flow, RealNVP):
            raise _lib.FabhipError("base_distribution must be a fab_torch_amd RealNVP for the HIP path "
                                   "(no generic / CPU fallback)")
        target = _owner(self.target_log_prob, _NativeTarget, "target_log_prob")
        return flow, target

    def run(self, batch_size: int, eps0=None, noise_a=None, noise_b=None, base_out=None):
        """Enqueue one AIS call; returns device tensors (point fields sized [batch_size], log_w, n_valid[2],
        stats[16]) without synchronising.  `base_out = (base_x [B, D], base_log_w [B])` additionally receives the
        chains' starting points and their log p - log q (generate_eval_data, ais.py:152-166)."""
        lib = _lib.load()
        flow, target = self._native_parts()
        op = self.transition_operator
        if bool(op.p_target) != bool(self.p_target) or (not self.p_target and op.alpha != self.alpha):
            # the reference keeps the two in sync through FABModel.set_ais_target (core.py:102-110)
            raise _lib.FabhipError("AIS and transition operator disagree on p_target / alpha")
        dev = flow._nf_model.q0.loc.device
        B, D, M = int(batch_size), flow.dim, self.n_intermediate_distributions
        hmc = isinstance(op, HamiltonianMonteCarlo)
        if not hmc and not isinstance(op, Metropolis):
            raise _lib.FabhipError("transition_operator must be a fab_torch_amd HamiltonianMonteCarlo / Metropolis")
        n_inner = op.n_outer if hmc else op.n_updates
        f32 = dict(dtype=torch.float32, device=dev)
        if eps0 is None:
            eps0 = torch.randn((B, D), **f32)
        if noise_a is None:
            noise_a = torch.randn((M, n_inner, B, D), **f32)
        if noise_b is None:
            noise_b = (torch.empty((M, n_inner, B), **f32).exponential_(1.0) if hmc
                       else torch.rand((M, n_inner, B), **f32))
        eps0, noise_a, noise_b = eps0.contiguous(), noise_a.contiguous(), noise_b.contiguous()
        assert noise_a.shape == (M, n_inner, B, D) and noise_b.shape == (M, n_inner, B)
        x = torch.empty((B, D), **f32)
        lq, lp, log_w = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32)
        gq = torch.empty((B, D), **f32) if hmc else None
        gp = torch.empty((B, D), **f32) if hmc else None
        n_valid = torch.zeros(2, dtype=torch.int32, device=dev)
        stats = torch.zeros(16, **f32)
        betas = (C.c_double * (M + 2))(*[float(b) for b in self.B_space])
        a = _lib.AisArgs()
        a.flow, _ = flow.native()
        a.target = target.native_target()
        a.B, a.M, a.betas = B, M, betas
        a.alpha = float(self.alpha) if self.alpha is not None else 0.0
        a.p_target = int(bool(self.p_target))
        a.transition = _lib.TRANSITION_HMC if hmc else _lib.TRANSITION_METROPOLIS
        a.eps0, a.noise_a, a.noise_b = eps0.data_ptr(), noise_a.data_ptr(), noise_b.data_ptr()
        if hmc:
            a.step_state, a.common_epsilon, a.mass = (op.epsilons.data_ptr(), op.common_epsilon.data_ptr(),
                                                      op.mass_vector.data_ptr())
            a.L, a.max_grad, a.target_p_accept = op.L, op.max_grad, op.target_p_accept
            a.tune = 0 if op.eval_mode else 1
        else:
            a.step_state, a.common_epsilon, a.mass = op.noise_scalings.data_ptr(), None, None
            a.L, a.max_grad, a.target_p_accept = 0, 0.0, op.target_prob_accept
            a.tune = 1 if (op.adjust_step_size and not op.eval_mode) else 0
        a.n_inner = n_inner
        a.point = _lib.Point(x.data_ptr(), lq.data_ptr(), lp.data_ptr(), gq.data_ptr() if hmc else None,
                             gp.data_ptr() if hmc else None)
        a.log_w, a.n_valid, a.stats = log_w.data_ptr(), n_valid.data_ptr(), stats.data_ptr()
        if hmc:          # per-outer-loop logging slots of the first / last distribution (hmc.py:173-183)
            a.p_accept_first, a.p_accept_last = op._p_accept_first.data_ptr(), op._p_accept_last.data_ptr()
            a.avg_distance_first, a.avg_distance_last = op._dist_first.data_ptr(), op._dist_last.data_ptr()
        if base_out is not None:
            bx, blw = base_out
            assert bx.shape == (B, D) and blw.shape == (B,) and bx.is_contiguous() and blw.is_contiguous()
            a.base_x, a.base_log_w = bx.data_ptr(), blw.data_ptr()
        nb = lib.fabhip_ais_workspace_bytes(B, D, n_inner)
        ws = self._ws.get(nb, dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nb
        _lib.check(lib.fabhip_ais_run(C.byref(a), _lib.stream_ptr()), "ais_run")
        return Point(x, lq, lp, gq, gp), log_w, n_valid, stats

    def sample_and_log_weights(self, batch_size: int, logging: bool = True, eps0=None, noise_a=None, noise_b=None
                               ) -> Tuple[Point, torch.Tensor]:
        point, log_w, n_valid, stats = self.run(batch_size, eps0, noise_a, noise_b)
        host = torch.cat([n_valid.float(), stats[:6]]).cpu()          # the single device->host read
        n_init, n_end = int(host[0]), int(host[1])
        if n_init == 0:
            raise Exception("No valid points generated in sampling the chain init")
        if n_end == 0:
            raise Exception("No valid points generated in sampling the chain end")
        if n_end != batch_size:
            print(f"{batch_size - n_end} nan/inf samples/log-probs/log-weights encountered.")
            point, log_w = point[:n_end], log_w[:n_end]
        st = host[2:]
        if logging:
            self._logging_info = LoggingInfo(ess_base=float(st[0]), ess_ais=float(st[3]), log_Z=float(st[4]))
        return point, log_w.detach()

    def generate_eval_data(self, outer_batch_size: int, inner_batch_size: int):
        """ais.py:132-188 — evaluation batches: the chains' starting points (flow samples) with log p - log q, and
        the AIS samples with their log-weights.  Everything stays on the device: the batches are enqueued back to
        back and the row counts of all of them are read with ONE device->host copy at the end (the reference
        concatenates on the CPU after a `.cpu()` per batch); the returned tensors live on the GPU."""
        flow, _ = self._native_parts()
        assert outer_batch_size % inner_batch_size == 0
        n_batches = outer_batch_size // inner_batch_size
        dev = flow._nf_model.q0.loc.device
        B, D = inner_batch_size, flow.dim
        base_x = torch.empty((n_batches, B, D), dtype=torch.float32, device=dev)
        base_lw = torch.empty((n_batches, B), dtype=torch.float32, device=dev)
        ais_x, ais_lw, counts = [], [], []
        for i in range(n_batches):
            point, log_w, n_valid, _ = self.run(B, base_out=(base_x[i], base_lw[i]))
            ais_x.append(point.x)
            ais_lw.append(log_w)
            counts.append(n_valid)
        host = torch.stack(counts).cpu()                       # the single synchronisation
        if int(host[:, 0].min()) == 0:
            raise Exception("No valid points generated in sampling the chain init")
        n0 = [int(v) for v in host[:, 0]]
        n1 = [int(v) if int(v) > 0 else n0[i] for i, v in enumerate(host[:, 1])]   # "chain end": print + keep (:170)
        if any(int(v) == 0 for v in host[:, 1]):
            print("No valid points generated in sampling the chain end")
        return (torch.cat([base_x[i, :n0[i]] for i in range(n_batches)]),
                torch.cat([base_lw[i, :n0[i]] for i in range(n_batches)]),
                torch.cat([ais_x[i][:n1[i]] for i in range(n_batches)]),
                torch.cat([ais_lw[i][:n1[i]] for i in range(n_batches)]))
