"""Oracle AIS loop, HMC and Metropolis transitions (PyTorch CPU, explicit noise inputs).

TEST INFRASTRUCTURE — never imported by the product package.

Follows, line by line in meaning (not in text):
* Point / create_point / grad_and_value ... fab/sampling_methods/base.py:7-72
* annealed log-density and its gradient ... fab/sampling_methods/base.py:76-118
  (incl. the reference quirk: the gradient hard-codes 2*beta on grad_log_p, base.py:116)
* HMC ................ fab/sampling_methods/transition_operators/hmc.py:90-202
* Metropolis ......... fab/sampling_methods/transition_operators/metropolis.py:51-74
  (incl. the stale ``x_prev_log_prob`` quirk, :53, and the inverted ``set_eval_mode``, :39-41)
* AIS driver ......... fab/sampling_methods/ais.py:53-105,108-129,190-213

Every random draw of the reference is an explicit argument here:
  eps0  [B, D]            base noise of the flow sample (normflows DiagGaussian)
  HMC:  noise_p [M, n_outer, B, D] ~ N(0,1) (torch.randn_like, hmc.py:134),
        noise_e [M, n_outer, B]    ~ Exp(1) (hmc.py:118)
  Metropolis: noise_x [M, n_updates, B, D] ~ N(0,1) (metropolis.py:57),
              noise_u [M, n_updates, B]    ~ U(0,1) (metropolis.py:65)
When NaN filtering shrinks the batch to B' rows the first B' rows of each noise slab are used.
"""
from typing import Callable, NamedTuple, Optional, Tuple

import numpy as np
import torch

from .numerical import effective_sample_size


class Point:
    def __init__(self, x, log_q, log_p, grad_log_q=None, grad_log_p=None):
        self.x, self.log_q, self.log_p = x, log_q, log_p
        self.grad_log_q, self.grad_log_p = grad_log_q, grad_log_p

    def __getitem__(self, idx):
        gq = self.grad_log_q[idx] if self.grad_log_q is not None else None
        gp = self.grad_log_p[idx] if self.grad_log_p is not None else None
        return Point(self.x[idx], self.log_q[idx], self.log_p[idx], gq, gp)

    def __setitem__(self, idx, v):
        self.x[idx] = v.x
        self.log_q[idx] = v.log_q
        self.log_p[idx] = v.log_p
        if self.grad_log_q is not None:
            self.grad_log_q[idx] = v.grad_log_q
            self.grad_log_p[idx] = v.grad_log_p

    def clone(self):
        c = lambda t: None if t is None else t.clone()
        return Point(c(self.x), c(self.log_q), c(self.log_p), c(self.grad_log_q), c(self.grad_log_p))


def grad_and_value(x, fn):
    x = x.detach().requires_grad_(True)
    y = fn(x)
    g = torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y))[0]
    return g.detach(), y.detach()


def create_point(x, log_q_fn, log_p_fn, with_grad: bool, log_q_x=None) -> Point:
    x = x.detach()
    if with_grad:
        gq, lq = grad_and_value(x, log_q_fn)       # a supplied log_q_x is ignored (base.py:65-68)
        gp, lp = grad_and_value(x, log_p_fn)
        return Point(x, lq, lp, gq, gp)
    lq = log_q_x if log_q_x is not None else log_q_fn(x)
    with torch.no_grad():
        lp = log_p_fn(x)
    return Point(x, lq.detach(), lp.detach())


def intermediate_log_prob(pt: Point, beta, alpha, p_target: bool):
    with torch.no_grad():
        if not p_target:
            return ((1 - beta) + beta * (1 - alpha)) * pt.log_q + beta * alpha * pt.log_p
        return (1 - beta) * pt.log_q + beta * pt.log_p


def grad_intermediate_log_prob(pt: Point, beta, alpha, p_target: bool):
    with torch.no_grad():
        if not p_target:
            return ((1 - beta) + beta * (1 - alpha)) * pt.grad_log_q + 2 * beta * pt.grad_log_p
        return (1 - beta) * pt.grad_log_q + beta * pt.grad_log_p


def beta_schedule(n_intermediate: int, spacing: str = "linear") -> torch.Tensor:
    """ais.py:108-129 — float64 tensor of length M+2."""
    if spacing == "geometric":
        n_lin = int(n_intermediate / 4)
        n_geo = n_intermediate - n_lin - 1
        b = np.concatenate([np.linspace(0, 0.01, n_lin + 2)[:-1], np.geomspace(0.01, 1, n_geo + 2)])
    elif spacing == "linear":
        b = np.linspace(0.0, 1.0, n_intermediate + 2)
    else:
        raise Exception(f"distribution spacing incorrectly specified: '{spacing}'")
    assert b.shape == (n_intermediate + 2,)
    return torch.tensor(b)


class HMC:
    def __init__(self, n_dist: int, dim: int, log_q_fn, log_p_fn, alpha=None, p_target=False,
                 epsilon=1.0, n_outer=1, L=5, mass_init=1.0, target_p_accept=0.65, max_grad=1e3,
                 common_epsilon_init_weight=0.1, eval_mode=False, dtype=torch.float32):
        self.n_dist, self.dim = n_dist, dim
        self.log_q_fn, self.log_p_fn = log_q_fn, log_p_fn
        self.alpha, self.p_target = alpha, p_target
        self.common_epsilon = torch.tensor([epsilon * common_epsilon_init_weight], dtype=dtype)
        self.epsilons = torch.ones([n_dist, n_outer], dtype=dtype) * epsilon * (1 - common_epsilon_init_weight)
        self.mass_vector = torch.ones(dim, dtype=dtype) * mass_init
        self.n_outer, self.L = n_outer, L
        self.target_p_accept, self.max_grad, self.eval_mode = target_p_accept, max_grad, eval_mode
        self.last_accept = None          # bool mask of the last outer step (for tests)
        self.last_margin = None
        self.last_p_accept = None

    uses_grad_info = True

    def _U(self, pt, beta):
        return -intermediate_log_prob(pt, beta, self.alpha, self.p_target)

    def _grad_U(self, pt, beta):
        g = -grad_intermediate_log_prob(pt, beta, self.alpha, self.p_target)
        return torch.nan_to_num(torch.clamp(g, max=self.max_grad, min=-self.max_grad),
                                nan=0.0, posinf=0.0, neginf=0.0)

    def transition(self, point: Point, i: int, beta, noise_p, noise_e) -> Point:
        """noise_p [n_outer, B, D], noise_e [n_outer, B]."""
        current = point
        for n in range(self.n_outer):
            eps = self.epsilons[i - 1, n] + self.common_epsilon
            p = noise_p[n, : point.x.shape[0]] * self.mass_vector
            current_p = p
            grad_u = self._grad_U(point, beta)
            for _ in range(self.L):
                p = p - eps * grad_u / 2
                x = point.x + eps / self.mass_vector * p
                point = create_point(x, self.log_q_fn, self.log_p_fn, with_grad=True)
                grad_u = self._grad_U(point, beta)
                p = p - eps * grad_u / 2
            lp_cur = -self._U(current, beta) - torch.sum(current_p ** 2 / self.mass_vector, -1) / 2
            lp_prop = -self._U(point, beta) - torch.sum(p ** 2 / self.mass_vector, -1) / 2
            log_acc = lp_prop - lp_cur
            # distance of the accept decision from its threshold (tests: a chain may only differ from this oracle
            # through a decision that sits within rounding of the threshold)
            self.last_margin = (log_acc + noise_e[n, : log_acc.shape[0]]).detach()
            valid = torch.isfinite(log_acc)
            log_acc = torch.nan_to_num(log_acc, nan=-float("inf"), posinf=-float("inf"),
                                       neginf=-float("inf"))
            accept = (log_acc > -noise_e[n, : log_acc.shape[0]]) & valid
            log_acc = torch.clamp(log_acc, max=0.0)
            log_p_accept_mean = torch.logsumexp(log_acc, -1) - torch.log(torch.tensor(log_acc.shape[0]))
            current[accept] = point[accept]
            self.last_accept, self.last_p_accept = accept, torch.exp(log_p_accept_mean)
            if not self.eval_mode:
                if log_p_accept_mean > torch.log(torch.tensor(self.target_p_accept)):
                    self.epsilons[i - 1, n] = self.epsilons[i - 1, n] * 1.05
                    self.common_epsilon = self.common_epsilon * 1.02
                else:
                    self.epsilons[i - 1, n] = self.epsilons[i - 1, n] / 1.05
                    self.common_epsilon = self.common_epsilon / 1.02
        return current


class Metropolis:
    def __init__(self, n_dist: int, dim: int, log_q_fn, log_p_fn, n_updates, alpha=None,
                 p_target=False, max_step_size=1.0, min_step_size=0.1, adjust_step_size=True,
                 target_p_accept=0.65, eval_mode=False, dtype=torch.float32):
        self.n_dist, self.dim, self.n_updates = n_dist, dim, n_updates
        self.log_q_fn, self.log_p_fn = log_q_fn, log_p_fn
        self.alpha, self.p_target = alpha, p_target
        self.adjust_step_size = adjust_step_size
        self.noise_scalings = torch.linspace(max_step_size, min_step_size, n_updates,
                                             dtype=dtype).repeat((n_dist, 1))
        self.target_prob_accept, self.eval_mode = target_p_accept, eval_mode
        self.last_accept = None

    uses_grad_info = False

    def transition(self, point: Point, i: int, beta, noise_x, noise_u) -> Point:
        """noise_x [n_updates, B, D], noise_u [n_updates, B]."""
        x_prev_log_prob = intermediate_log_prob(point, beta, self.alpha, self.p_target)  # never refreshed
        for n in range(self.n_updates):
            B = point.x.shape[0]
            x_prop = point.x + noise_x[n, :B] * self.noise_scalings[i - 1, n]
            prop = create_point(x_prop, self.log_q_fn, self.log_p_fn, with_grad=False)
            prop_lp = intermediate_log_prob(prop, beta, self.alpha, self.p_target)
            acc = torch.exp(prop_lp - x_prev_log_prob)
            acc = torch.nan_to_num(acc, nan=0.0, posinf=0.0, neginf=0.0)
            accept = acc > noise_u[n, :B]
            point[accept] = prop[accept]
            self.last_accept = accept
            if self.adjust_step_size and not self.eval_mode:
                p_accept = torch.mean(torch.clamp_max(acc, 1))
                if p_accept > self.target_prob_accept:
                    self.noise_scalings[i - 1, n] = self.noise_scalings[i - 1, n] * 1.05
                else:
                    self.noise_scalings[i - 1, n] = self.noise_scalings[i - 1, n] / 1.05
        return point


class LoggingInfo(NamedTuple):
    ess_base: float
    ess_ais: float
    log_Z: float


def remove_nan_and_infs(point: Point, log_w, descriptor="chain init", raise_exception=True):
    valid = ~torch.isinf(point.log_p) & ~torch.isnan(point.log_p) & \
            ~torch.isinf(point.log_q) & ~torch.isnan(point.log_q)
    if torch.sum(valid) == 0:
        if raise_exception:
            raise Exception(f"No valid points generated in sampling the {descriptor}")
        return point, log_w
    return point[valid], log_w[valid]


class AIS:
    def __init__(self, sample_eps_fn: Callable, log_q_fn, log_p_fn, transition_operator, p_target: bool,
                 alpha: Optional[float] = None, n_intermediate_distributions: int = 1,
                 distribution_spacing_type: str = "linear"):
        if not p_target:
            assert alpha is not None
        self.sample_eps_fn, self.log_q_fn, self.log_p_fn = sample_eps_fn, log_q_fn, log_p_fn
        self.transition_operator = transition_operator
        self.p_target, self.alpha = p_target, alpha
        self.M = n_intermediate_distributions
        self.B_space = beta_schedule(n_intermediate_distributions, distribution_spacing_type)
        self.snapshots = None

    def sample_and_log_weights(self, eps0, noise_a, noise_b, keep_snapshots=False
                               ) -> Tuple[Point, torch.Tensor, LoggingInfo]:
        batch_size = eps0.shape[0]
        x, log_q0 = self.sample_eps_fn(eps0)
        point = create_point(x, self.log_q_fn, self.log_p_fn,
                             with_grad=self.transition_operator.uses_grad_info, log_q_x=log_q0)
        log_w = intermediate_log_prob(point, self.B_space[1], self.alpha, self.p_target) - log_q0
        log_w = log_w.detach()
        point, log_w = remove_nan_and_infs(point, log_w, "chain init")
        with torch.no_grad():
            ess_base = effective_sample_size(point.log_p - point.log_q).item()
        snaps = [(point.clone(), log_w.clone())] if keep_snapshots else None
        self.margins = [None]                 # per transition: accept-decision margins of the last outer step
        for j in range(1, self.M + 1):
            point = self.transition_operator.transition(point, j, self.B_space[j], noise_a[j - 1], noise_b[j - 1])
            if self.B_space[j + 1] != self.B_space[j]:
                num = intermediate_log_prob(point, self.B_space[j + 1], self.alpha, self.p_target)
                den = intermediate_log_prob(point, self.B_space[j], self.alpha, self.p_target)
                log_w = log_w + (num - den)
            if keep_snapshots:
                snaps.append((point.clone(), log_w.clone()))
                self.margins.append(getattr(self.transition_operator, "last_margin", None))
        point, log_w = remove_nan_and_infs(point, log_w, "chain end")
        with torch.no_grad():
            ess_ais = effective_sample_size(log_w).item()
            lz = torch.logsumexp(log_w, dim=0)
            log_Z = (lz - torch.log(torch.ones_like(lz) * batch_size)).item()
        self.snapshots = snaps
        return point, log_w.detach(), LoggingInfo(ess_base, ess_ais, log_Z)
