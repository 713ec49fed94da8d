"""HMC and Metropolis transition operators with the reference's plug-in interface
(fab/sampling_methods/transition_operators/base.py:12-85, hmc.py:8-202, metropolis.py:9-74):
same constructor arguments, `transition(point, i, beta) -> Point` (mutating the caller's Point like the
reference), `uses_grad_info`, `get_logging_info()`, `set_eval_mode()`, state-dict buffers
`common_epsilon[1]`, `epsilons[M, n_outer]`, `mass_vector[D]` / `noise_scalings[M, n_updates]`.

A transition is ONE C-ABI call (fabhip_hmc_transition / fabhip_metropolis_transition): all leapfrogs,
flow + target evaluations, accept/reject, commit and step-size adaptation run on the GPU; the step-size
state lives in the registered device buffers and is updated in place (no host synchronisation).
"""
import ctypes as C
from typing import Dict, Optional, Union

import torch

from . import _lib
from .flow import RealNVP
from .point import Point
from .targets import _NativeTarget


def anneal_coefs(beta, alpha, p_target) -> _lib.Anneal:
    a = _lib.Anneal()
    _lib.load().fabhip_anneal_coefs(float(beta), float(alpha if alpha is not None else 0.0), int(bool(p_target)),
                                    C.byref(a))
    return a


def _owner(fn, cls, what):
    obj = getattr(fn, "__self__", None)
    if not isinstance(obj, cls):
        raise _lib.FabhipError(
            f"{what} must be the bound `log_prob` of a fab_torch_amd {cls.__name__} for the HIP path "
            f"(got {fn!r}); there is no CPU / generic-callable fallback")
    return obj


class TransitionOperator(torch.nn.Module):
    def __init__(self, n_ais_intermediate_distributions: int, dim: int, base_log_prob, target_log_prob,
                 p_target: bool = True, alpha: float = None):
        self.dim = dim
        self.target_log_prob = target_log_prob
        self.base_log_prob = base_log_prob
        self.alpha = alpha
        self.n_ais_intermediate_distributions = n_ais_intermediate_distributions
        self.p_target = p_target
        super().__init__()
        self._ws = _lib.Workspace()

    # resolved lazily so that the operator can be constructed before `.cuda()`
    @property
    def flow(self) -> RealNVP:
        return _owner(self.base_log_prob, RealNVP, "base_log_prob")

    @property
    def target(self) -> _NativeTarget:
        return _owner(self.target_log_prob, _NativeTarget, "target_log_prob")

    def create_new_point(self, x: torch.Tensor) -> Point:
        return create_point(x, self.flow, self.target, with_grad=self.uses_grad_info)

    @property
    def uses_grad_info(self) -> bool:
        raise NotImplementedError

    def get_logging_info(self):
        raise NotImplementedError

    def transition(self, point: Point, i: int, beta: float) -> Point:
        raise NotImplementedError

    def set_eval_mode(self, eval_setting: bool):
        raise NotImplementedError


def create_point(x: torch.Tensor, flow: RealNVP, target: _NativeTarget, with_grad: bool,
                 log_q_x: Optional[torch.Tensor] = None) -> Point:
    """fab/sampling_methods/base.py:59-72 on the GPU (one fused launch)."""
    lib = _lib.load()
    _lib.require_device(x, "x")
    x = x.detach().contiguous().float()
    B, D = x.shape
    f, _ = flow.native()
    t = target.native_target()
    lq = torch.empty(B, dtype=torch.float32, device=x.device)
    lp = torch.empty_like(lq)
    gq = torch.empty_like(x) if with_grad else None
    gp = torch.empty_like(x) if with_grad else None
    p = _lib.Point(x.data_ptr(), lq.data_ptr(), lp.data_ptr(), gq.data_ptr() if with_grad else None,
                   gp.data_ptr() if with_grad else None)
    _lib.check(lib.fabhip_create_point(C.byref(f), C.byref(t), C.byref(p), int(with_grad), B, _lib.stream_ptr()),
               "create_point")
    if not with_grad and log_q_x is not None:
        lq = log_q_x.detach()
    return Point(x, lq, lp, gq, gp)


def _point_struct(point: Point, with_grad: bool) -> _lib.Point:
    for t in (point.x, point.log_q, point.log_p):
        _lib.require_device(t, "point")
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise _lib.FabhipError("Point tensors must be contiguous float32 (they are updated in place)")
    gq = point.grad_log_q.data_ptr() if with_grad else None
    gp = point.grad_log_p.data_ptr() if with_grad else None
    return _lib.Point(point.x.data_ptr(), point.log_q.data_ptr(), point.log_p.data_ptr(), gq, gp)


class HamiltonianMonteCarlo(TransitionOperator):
    def __init__(self, n_ais_intermediate_distributions: int, dim: int, base_log_prob, target_log_prob,
                 alpha: float = None, p_target: bool = False, epsilon: float = 1.0, n_outer: int = 1, L: int = 5,
                 mass_init: Union[float, torch.Tensor] = 1.0, target_p_accept: float = 0.65,
                 max_grad: float = 1e3, tune_period: bool = False, common_epsilon_init_weight: float = 0.1,
                 eval_mode: bool = False):
        super().__init__(n_ais_intermediate_distributions, dim, base_log_prob, target_log_prob, alpha=alpha,
                         p_target=p_target)
        if isinstance(mass_init, torch.Tensor):
            assert mass_init.shape == (dim,)
        self.tune_period = tune_period
        self.register_buffer("common_epsilon", torch.tensor([epsilon * common_epsilon_init_weight]))
        self.register_buffer("epsilons", torch.ones([n_ais_intermediate_distributions, n_outer]) * epsilon *
                             (1 - common_epsilon_init_weight))
        self.register_buffer("mass_vector", torch.ones(dim) * mass_init)
        self.n_outer, self.L = n_outer, L
        self.target_p_accept, self.max_grad = target_p_accept, max_grad
        self.eval_mode = eval_mode
        # logging state (device tensors; read lazily by get_logging_info)
        self.register_buffer("_p_accept_first", torch.zeros(n_outer), persistent=False)
        self.register_buffer("_p_accept_last", torch.zeros(n_outer), persistent=False)
        self.register_buffer("_dist_first", torch.zeros(1), persistent=False)
        self.register_buffer("_dist_last", torch.zeros(1), persistent=False)

    @property
    def uses_grad_info(self) -> bool:
        return True

    def set_eval_mode(self, eval_setting: bool):
        self.eval_mode = eval_setting

    def get_epsilon(self, i: int, n: int) -> torch.Tensor:
        return self.epsilons[i - 1, n] + self.common_epsilon

    def get_logging_info(self) -> dict:
        """Keys of hmc.py:59-88; ONE device->host read for all of them."""
        M, no = self.n_ais_intermediate_distributions, self.n_outer
        eps_first = self.epsilons[-1, 0] + self.common_epsilon        # the reference's get_epsilon(0, 0): index -1
        eps_last = self.epsilons[M - 2, 0] + self.common_epsilon if M > 1 else eps_first
        host = torch.cat([self._p_accept_first, self._p_accept_last, self._dist_first, self._dist_last,
                          eps_first.reshape(1), eps_last.reshape(1)]).tolist()
        d = {}
        for n in range(no):
            d[f"dist0_p_accept_{n}"] = host[n]
        if M > 1:
            for n in range(no):
                d[f"dist{M - 1}_p_accept_{n}"] = host[no + n]
        d["epsilons_dist0_loop0"] = host[2 * no + 2]
        if M > 1:
            d[f"epsilons_dist{M - 1}_loop0"] = host[2 * no + 3]
        d["average_distance_dist0"] = host[2 * no]
        if M > 1:
            d[f"average_distance_dist_{M - 1}"] = host[2 * no + 1]
        return d

    def transition(self, point: Point, i: int, beta: float, log_w: torch.Tensor = None, beta_next=None,
                   noise_p: torch.Tensor = None, noise_e: torch.Tensor = None) -> Point:
        """`noise_p [n_outer, B, D]` / `noise_e [n_outer, B]` may be supplied (parity tests); otherwise they are
        drawn from the device generator.  With `log_w`/`beta_next` the AIS increment (ais.py:93-100) is fused."""
        lib = _lib.load()
        B, D = point.x.shape
        dev = point.x.device
        if noise_p is None:
            noise_p = torch.randn((self.n_outer, B, D), dtype=torch.float32, device=dev)
        if noise_e is None:
            noise_e = torch.empty((self.n_outer, B), dtype=torch.float32, device=dev).exponential_(1.0)
        noise_p, noise_e = noise_p.contiguous(), noise_e.contiguous()
        a = _lib.HmcArgs()
        a.flow, _ = self.flow.native()
        a.target = self.target.native_target()
        a.point = _point_struct(point, True)
        a.B, a.n_valid = B, None
        a.cur = anneal_coefs(beta, self.alpha, self.p_target)
        a.next = anneal_coefs(beta_next if beta_next is not None else beta, self.alpha, self.p_target)
        a.log_w = log_w.data_ptr() if log_w is not None else None
        a.noise_p, a.noise_e = noise_p.data_ptr(), noise_e.data_ptr()
        a.epsilons = self.epsilons.data_ptr() + 4 * (i - 1) * self.n_outer
        a.common_epsilon, a.mass = self.common_epsilon.data_ptr(), self.mass_vector.data_ptr()
        a.n_outer, a.L = self.n_outer, self.L
        a.max_grad, a.target_p_accept = self.max_grad, self.target_p_accept
        a.tune = 0 if self.eval_mode else 1
        M = self.n_ais_intermediate_distributions
        a.p_accept, a.avg_distance = None, None
        if i == 1:
            a.p_accept, a.avg_distance = self._p_accept_first.data_ptr(), self._dist_first.data_ptr()
        elif i == M:
            a.p_accept, a.avg_distance = self._p_accept_last.data_ptr(), self._dist_last.data_ptr()
        nb = lib.fabhip_hmc_workspace_bytes(B, D, self.n_outer)
        ws = self._ws.get(nb, dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nb
        _lib.check(lib.fabhip_hmc_transition(C.byref(a), _lib.stream_ptr()), "hmc_transition")
        return point


class Metropolis(TransitionOperator):
    def __init__(self, n_ais_intermediate_distributions: int, dim: int, base_log_prob, target_log_prob,
                 n_updates, alpha: float = None, p_target: bool = False, max_step_size=1.0, min_step_size=0.1,
                 adjust_step_size=True, target_p_accept=0.65, eval_mode: bool = False):
        super().__init__(n_ais_intermediate_distributions, dim, base_log_prob, target_log_prob, alpha=alpha,
                         p_target=p_target)
        self.n_distributions = n_ais_intermediate_distributions
        self.n_updates = n_updates
        self.adjust_step_size = adjust_step_size
        self.register_buffer("noise_scalings", torch.linspace(max_step_size, min_step_size, n_updates).repeat(
            (n_ais_intermediate_distributions, 1)))
        self.target_prob_accept = target_p_accept
        self.eval_mode = eval_mode

    @property
    def uses_grad_info(self) -> bool:
        return False

    def set_eval_mode(self, eval_setting: bool):
        # NB the reference inverts the flag here (metropolis.py:39-41); kept for drop-in behaviour.
        self.eval_mode = not eval_setting

    def get_logging_info(self) -> Dict:
        first, last = self.noise_scalings[0, [0, -1]].tolist()             # one device->host read
        return {"noise_scaling_0_0": first, "noise_scaling_0_-1": last}

    def transition(self, point: Point, i: int, beta: float, log_w: torch.Tensor = None, beta_next=None,
                   noise_x: torch.Tensor = None, noise_u: torch.Tensor = None) -> Point:
        lib = _lib.load()
        B, D = point.x.shape
        dev = point.x.device
        if noise_x is None:
            noise_x = torch.randn((self.n_updates, B, D), dtype=torch.float32, device=dev)
        if noise_u is None:
            noise_u = torch.rand((self.n_updates, B), dtype=torch.float32, device=dev)
        noise_x, noise_u = noise_x.contiguous(), noise_u.contiguous()
        a = _lib.MetropolisArgs()
        a.flow, _ = self.flow.native()
        a.target = self.target.native_target()
        a.point = _point_struct(point, False)
        a.B, a.n_valid = B, None
        a.cur = anneal_coefs(beta, self.alpha, self.p_target)
        a.next = anneal_coefs(beta_next if beta_next is not None else beta, self.alpha, self.p_target)
        a.log_w = log_w.data_ptr() if log_w is not None else None
        a.noise_x, a.noise_u = noise_x.data_ptr(), noise_u.data_ptr()
        a.noise_scalings = self.noise_scalings.data_ptr() + 4 * (i - 1) * self.n_updates
        a.n_updates = self.n_updates
        a.target_p_accept = self.target_prob_accept
        a.tune = 1 if (self.adjust_step_size and not self.eval_mode) else 0
        nb = lib.fabhip_metropolis_workspace_bytes(B, D, self.n_updates)
        ws = self._ws.get(nb, dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nb
        _lib.check(lib.fabhip_metropolis_transition(C.byref(a), _lib.stream_ptr()), "metropolis_transition")
        return point
