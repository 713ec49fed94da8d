"""HMC and Metropolis transition operators with the reference's plug-in interface
(fab/sampling_methods/transition_operators/base.py:12-85, hmc.py:8-202, metropolis.py:9-74):
same constructor arguments, `transition(point, i, beta) -> Point` (mutating the caller's Point like the
reference), `uses_grad_info`, `get_logging_info()`, `set_eval_mode()`, state-dict buffers
`common_epsilon[1]`, `epsilons[M, n_outer]`, `mass_vector[D]` / `noise_scalings[M, n_updates]`.

A transition is ONE custom-op call (torch.ops.fabhip.hmc_transition / metropolis_transition -> the C ABI): all leapfrogs,
flow + target evaluations, accept/reject, commit and step-size adaptation run on the GPU; the step-size
state lives in the registered device buffers and is updated in place (no host synchronisation).
"""
import os
from typing import Dict, Optional, Union

import torch

from . import _ops
from .flow import RealNVP
from .point import Point
from .targets import _NativeTarget


def anneal_coefs(beta, alpha, p_target):
    """(c_q, c_p, g_q, g_p) of fab/sampling_methods/base.py:76-118 as float32 values (fabhip_anneal_coefs)."""
    return tuple(_ops.load().anneal_coefs(float(beta), float(alpha if alpha is not None else 0.0), bool(p_target)))


def get_intermediate_log_prob(x: Point, beta: float, alpha: Optional[float], p_target: bool) -> torch.Tensor:
    """fab/sampling_methods/base.py:76-97: log of the annealed density at a Point (fabhip_anneal_log_prob)."""
    return _ops.load().anneal_log_prob(x.log_q.contiguous(), x.log_p.contiguous(), float(beta),
                                       float(alpha if alpha is not None else 0.0), bool(p_target))


def get_grad_intermediate_log_prob(x: Point, beta: float, alpha: Optional[float], p_target: bool) -> torch.Tensor:
    """fab/sampling_methods/base.py:100-118, incl. its hard-coded 2 beta on grad log p."""
    _, _, g_q, g_p = anneal_coefs(beta, alpha, p_target)
    return g_q * x.grad_log_q + g_p * x.grad_log_p


def _owner(fn, cls, what):
    obj = getattr(fn, "__self__", None)
    if not isinstance(obj, cls):
        raise _ops.FabhipError(
            f"{what} must be the bound `log_prob` of a fab_torch_amd {cls.__name__} for the fused HIP path "
            f"(got {fn!r})")
    return obj


def _owner_or_none(fn, cls):
    obj = getattr(fn, "__self__", None)
    return obj if isinstance(obj, cls) else None


def grad_and_value(x: torch.Tensor, forward_fn):
    """fab/sampling_methods/base.py:50-56: value and gradient w.r.t. x of a plug-in's log-density (its own code).
    A `log_prob` bound to one of this package's HIP-backed distributions returns both from one kernel sweep
    (`log_prob_and_grad`) without building an autograd graph (nor, for a flow being trained, a parameter tape)."""
    owner = getattr(forward_fn, "__self__", None)
    if owner is not None and getattr(forward_fn, "__name__", "") == "log_prob" and \
            getattr(type(owner), "__module__", "").startswith(__package__ + ".") and hasattr(owner, "log_prob_and_grad"):
        y, grad = owner.log_prob_and_grad(x.detach())
        return grad, y
    x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        y = forward_fn(x)
        grad = torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y))[0]
    return grad.detach(), y.detach()


def create_point_generic(x: torch.Tensor, log_q_fn, log_p_fn, with_grad: bool,
                         log_q_x: Optional[torch.Tensor] = None) -> Point:
    """create_point for ANY plug-in callables (base.py:59-72): the densities are evaluated by the plug-ins themselves
    (with autograd for the gradients), the result is laid out as the contiguous float32 Point the HIP kernels update."""
    _ops.require_device(x, "x")
    c = lambda t: t.detach().contiguous().float()          # noqa: E731
    x = c(x)
    if with_grad:
        gq, lq = grad_and_value(x, log_q_fn)                 # a supplied log_q_x is ignored here, like the reference
        gp, lp = grad_and_value(x, log_p_fn)
        return Point(x, c(lq), c(lp), c(gq), c(gp))
    with torch.no_grad():
        lq = log_q_x if log_q_x is not None else log_q_fn(x)
        lp = log_p_fn(x)
    return Point(x, c(lq), c(lp))


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

    # resolved lazily so that the operator can be constructed before `.cuda()`
    @property
    def flow(self) -> RealNVP:
        return _owner(self.base_log_prob, RealNVP, "base_log_prob")

    @property
    def target(self) -> _NativeTarget:
        return _owner(self.target_log_prob, _NativeTarget, "target_log_prob")

    @property
    def is_native(self) -> bool:
        """Both plug-ins are fabhip-native: the fused kernels evaluate the densities themselves.  Otherwise the
        generic path runs: densities by the plug-ins' own code, the rest of the transition as HIP elementwise kernels."""
        return (_owner_or_none(self.base_log_prob, RealNVP) is not None
                and _owner_or_none(self.target_log_prob, _NativeTarget) is not None)

    def intermediate_target_log_prob(self, point: Point, beta: float) -> torch.Tensor:
        """transition_operators/base.py:37-43 (fabhip_anneal_log_prob: the coefficients of base.py:76-97)."""
        a = float(self.alpha if self.alpha is not None else 0.0)
        return _ops.load().anneal_log_prob(point.log_q.contiguous(), point.log_p.contiguous(), float(beta), a,
                                           bool(self.p_target))

    def grad_intermediate_target_log_prob(self, point: Point, beta: float) -> torch.Tensor:
        """transition_operators/base.py:45-54, incl. the reference's 2 beta on grad log p (base.py:116)."""
        c_q, c_p, g_q, g_p = anneal_coefs(beta, self.alpha, self.p_target)
        return g_q * point.grad_log_q + g_p * point.grad_log_p

    def create_new_point(self, x: torch.Tensor) -> Point:
        if self.is_native:
            return create_point(x, self.flow, self.target, with_grad=self.uses_grad_info)
        return create_point_generic(x, self.base_log_prob, self.target_log_prob, with_grad=self.uses_grad_info)

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
    _ops.require_device(x, "x")
    x = x.detach().contiguous().float()
    lq, lp, gq, gp = _ops.load().create_point(*flow.native(), *target.native_target(), x, bool(with_grad),
                                              _ops.precision_of(flow))
    if not with_grad and log_q_x is not None:
        lq = log_q_x.detach()
    return Point(x, lq, lp, gq if with_grad else None, gp if with_grad else None)


def _check_point(point: Point, with_grad: bool):
    ts = [point.x, point.log_q, point.log_p] + ([point.grad_log_q, point.grad_log_p] if with_grad else [])
    for t in ts:
        _ops.require_device(t, "point")
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise _ops.FabhipError("Point tensors must be contiguous float32 (they are updated in place)")


class HamiltonianMonteCarlo(TransitionOperator):
    force_stepwise = False      # tests only: run the spline flow through the step-by-step generic path

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
        B, D = point.x.shape
        dev = point.x.device
        if noise_p is None:
            noise_p = torch.randn((self.n_outer, B, D), dtype=torch.float32, device=dev)
        if noise_e is None:
            noise_e = torch.empty((self.n_outer, B), dtype=torch.float32, device=dev).exponential_(1.0)
        _check_point(point, True)
        M = self.n_ais_intermediate_distributions
        p_accept, avg_distance = None, None
        if i == 1:
            p_accept, avg_distance = self._p_accept_first, self._dist_first
        elif i == M:
            p_accept, avg_distance = self._p_accept_last, self._dist_last
        if not self.is_native:
            spline = self._spline_parts()
            if spline is not None:                     # spline flow + native target: the same steps, enqueued by one op
                flow, target = spline
                _ops.load().spline_hmc_transition(
                    *flow.native(), *target.native_target(), point.x, point.log_q, point.log_p, point.grad_log_q,
                    point.grad_log_p, log_w, float(beta), float(beta_next if beta_next is not None else beta),
                    float(self.alpha if self.alpha is not None else 0.0), bool(self.p_target), noise_p.contiguous(),
                    noise_e.contiguous(), self.epsilons[i - 1], self.common_epsilon, self.mass_vector, self.n_outer, self.L,
                    float(self.max_grad), float(self.target_p_accept), not self.eval_mode, p_accept, avg_distance,
                    _ops.precision_of(flow))
                return point
            return self._transition_generic(point, i, beta, log_w, beta_next, noise_p.contiguous(),
                                            noise_e.contiguous(), p_accept, avg_distance)
        _ops.load().hmc_transition(
            *self.flow.native(), *self.target.native_target(), point.x, point.log_q, point.log_p, point.grad_log_q,
            point.grad_log_p, log_w, float(beta), float(beta_next if beta_next is not None else beta),
            float(self.alpha if self.alpha is not None else 0.0), bool(self.p_target), noise_p.contiguous(),
            noise_e.contiguous(), self.epsilons[i - 1], self.common_epsilon, self.mass_vector, self.L,
            float(self.max_grad), float(self.target_p_accept), not self.eval_mode, p_accept, avg_distance,
            _ops.precision_of(self.flow))
        return point


    def _spline_parts(self):
        """(spline flow, native target) when the base distribution is this package's spline flow and the target is
        native - the host-fused transition op applies; else None (step-by-step generic path)."""
        if self.force_stepwise:                         # tests: force the step-by-step generic path
            return None
        from .spline_flow import CircularCoupledRQSFlow
        flow = _owner_or_none(self.base_log_prob, CircularCoupledRQSFlow)
        target = _owner_or_none(self.target_log_prob, _NativeTarget)
        return (flow, target) if flow is not None and target is not None else None

    def save_model(self, save_path, epoch=None):
        """hmc.py:204-214 (state dict under HMC_model[_epoch{n}] + a text description)."""
        import os
        tag = "" if epoch is None else f"_epoch{epoch}"
        with open(os.path.join(str(save_path), f"HMC_model_info{tag}.txt"), "w") as g:
            g.write(str(dict(n_distributions=self.n_ais_intermediate_distributions, n_outer=self.n_outer, L=self.L,
                             dim=self.dim, target_p_accept=self.target_p_accept, tune_period=self.tune_period)))
        torch.save(self.state_dict(), os.path.join(str(save_path), f"HMC_model{tag}"))

    def load_model(self, save_path, epoch=None, device="cpu"):
        """hmc.py:216-222."""
        import os
        tag = "" if epoch is None else f"_epoch{epoch}"
        self.load_state_dict(torch.load(os.path.join(str(save_path), f"HMC_model{tag}"), map_location=torch.device(device)))
        print("loaded HMC model")

    def _transition_generic(self, point, i, beta, log_w, beta_next, noise_p, noise_e, p_accept, avg_distance):
        """hmc.py:129-160 for arbitrary plug-ins: the L x n_outer density evaluations are the plug-ins' own code, the
        momentum / position updates, the Metropolis test, the in-place commit, the log-weight increment and the
        step-size adaptation are fabhip elementwise kernels sharing one workspace (include/fabhip.h, generic path)."""
        ops = _ops.load()
        assert self.L >= 1
        B, D = point.x.shape
        alpha = float(self.alpha if self.alpha is not None else 0.0)
        beta, beta_next = float(beta), float(beta_next if beta_next is not None else beta)
        ws = ops.generic_workspace(point.x, B, D)
        start = point
        for n in range(self.n_outer):
            eps_n = self.epsilons[i - 1, n:n + 1]
            ops.hmc_generic_begin(start.x, start.grad_log_q, start.grad_log_p, point.log_q, point.log_p, beta, alpha,
                                  bool(self.p_target), noise_p[n], self.mass_vector, float(self.max_grad), ws)
            prop = None
            for _ in range(self.L):
                x_new = ops.hmc_generic_leap_pre(B, D, eps_n, self.common_epsilon, self.mass_vector, ws)
                prop = self.create_new_point(x_new)
                ops.hmc_generic_leap_post(prop.grad_log_q, prop.grad_log_p, beta, alpha, bool(self.p_target),
                                          float(self.max_grad), eps_n, self.common_epsilon, ws)
            last = n + 1 == self.n_outer
            ops.hmc_generic_accept(prop.log_q, prop.log_p, prop.grad_log_q, prop.grad_log_p, point.x, point.log_q,
                                   point.log_p, point.grad_log_q, point.grad_log_p, log_w if last else None, beta,
                                   beta_next, alpha, bool(self.p_target), noise_e[n], self.mass_vector, eps_n,
                                   self.common_epsilon, float(self.target_p_accept), not self.eval_mode,
                                   p_accept[n:n + 1] if p_accept is not None else None, avg_distance, ws)
            start = prop                       # the reference continues from the PROPOSAL (hmc.py:133-142)
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
        B, D = point.x.shape
        dev = point.x.device
        if noise_x is None:
            noise_x = torch.randn((self.n_updates, B, D), dtype=torch.float32, device=dev)
        if noise_u is None:
            noise_u = torch.rand((self.n_updates, B), dtype=torch.float32, device=dev)
        _check_point(point, False)
        if not self.is_native:
            return self._transition_generic(point, i, beta, log_w, beta_next, noise_x.contiguous(), noise_u.contiguous())
        _ops.load().metropolis_transition(
            *self.flow.native(), *self.target.native_target(), point.x, point.log_q, point.log_p, log_w, float(beta),
            float(beta_next if beta_next is not None else beta), float(self.alpha if self.alpha is not None else 0.0),
            bool(self.p_target), noise_x.contiguous(), noise_u.contiguous(), self.noise_scalings[i - 1],
            float(self.target_prob_accept), bool(self.adjust_step_size and not self.eval_mode))
        return point

    def _transition_generic(self, point, i, beta, log_w, beta_next, noise_x, noise_u):
        """metropolis.py:51-74 for arbitrary plug-ins (densities by the plug-ins, propose / accept / commit / step
        adaptation as fabhip elementwise kernels)."""
        ops = _ops.load()
        alpha = float(self.alpha if self.alpha is not None else 0.0)
        beta, beta_next = float(beta), float(beta_next if beta_next is not None else beta)
        prev = ops.anneal_log_prob(point.log_q, point.log_p, beta, alpha, bool(self.p_target))   # never refreshed (:53)
        tune = bool(self.adjust_step_size and not self.eval_mode)
        for n in range(self.n_updates):
            scale = self.noise_scalings[i - 1, n:n + 1]
            x_new = ops.metropolis_generic_propose(point.x, noise_x[n], scale)
            new = self.create_new_point(x_new)
            ops.metropolis_generic_accept(x_new, new.log_q, new.log_p, point.x, point.log_q, point.log_p, prev,
                                          noise_u[n], beta, alpha, bool(self.p_target), scale,
                                          float(self.target_prob_accept), tune)
        if log_w is not None:
            ops.log_w_update(point.log_q, point.log_p, beta, beta_next, alpha, bool(self.p_target), log_w)
        return point
