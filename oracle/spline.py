"""Oracle circular / linear-tail rational-quadratic spline coupling flow (PyTorch CPU) - TEST INFRASTRUCTURE.

What the reference builds for alanine dipeptide (experiments/make_flow/make_aldp_model.py:57-71,121-134,146-167 with
experiments/aldp/config/fab_buff.yaml:20-36): 12 x `nf.flows.CircularCoupledRationalQuadraticSpline(60, blocks_per_layer=1,
hidden_units=256, ind_circ, tail_bound, num_bins=8, init_identity=True, mask=alternating random binary mask)`, a
`PeriodicShift` after every second layer, a final `PeriodicWrap`, base `UniformGaussian(60, ind_circ, scale)`.

All of these classes live in the third-party package `normflows` (requirements.txt:3, unpinned), which is ABSENT from
/root/reference and cannot be installed here, and the reference holds no vector at this boundary
-> **parity unpinned**.  The arithmetic below restates the published definitions:
  * monotone rational-quadratic splines: Durkan, Bekasov, Murray, Papamakarios, "Neural Spline Flows" (NeurIPS 2019),
    eqs. (4)-(8) and the nflows / normflows `rational_quadratic_spline` (softmax-normalised bin widths / heights with
    a 1e-3 floor, softplus + 1e-3 knot derivatives, K + 1 knots, closed-form inverse through the quadratic root);
  * tails: identity outside [-B, B] with boundary derivative 1 ("linear"), or periodic with the first and last knot
    derivative tied ("circular"; Rezende et al., "Normalizing Flows on Tori and Spheres", ICML 2020);
  * coupling: the masked-out ("identity") features condition a residual MLP (`ResidualNet`: Linear, one pre-activation
    residual block, Linear) whose circular inputs enter through learned `w0 sin(s x) + w1 cos(s x)` features; it emits
    3K + 1 numbers per transformed feature (K widths, K heights - both divided by sqrt(hidden) - and K + 1 derivatives);
    the identity features themselves go through an unconditional element-wise spline
    (`apply_unconditional_transform=True`);
  * direction convention of normflows' wrapper: `flow.inverse` (used by log_prob) EVALUATES the spline, `flow.forward`
    (used by sampling) inverts it.
Module / parameter names follow normflows (`prqct.transform_net.{initial_layer,blocks.0.linear_layers.{0,1},
final_layer}`, `prqct.transform_net.preprocessing.weights`, `prqct.unconditional_transform.unnormalized_{widths,heights,
derivatives}`) so that a real normflows checkpoint of this architecture would load by key; whether every detail
(e.g. K + 1 free derivatives for list-valued tails) matches the installed normflows version cannot be verified offline.

Pinned by self-consistency only (tests/test_oracle_spline.py): inverse o forward = id, log-det = log|det J| of the
autograd Jacobian, log_prob(sample) = returned log q, periodicity of the circular coordinates."""
import math
from typing import List, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

MIN_BIN_WIDTH = 1e-3
MIN_BIN_HEIGHT = 1e-3
MIN_DERIVATIVE = 1e-3


def _searchsorted(bin_locations, inputs, eps=1e-6):
    bl = bin_locations.clone()
    bl[..., -1] += eps
    return torch.sum(inputs[..., None] >= bl, dim=-1) - 1


def rational_quadratic_spline(inputs, uw, uh, ud, inverse, left, right, bottom, top):
    """Monotone RQ spline on [left, right] -> [bottom, top]; uw/uh [..., K], ud [..., K + 1] unnormalised."""
    K = uw.shape[-1]
    widths = F.softmax(uw, dim=-1)
    widths = MIN_BIN_WIDTH + (1 - MIN_BIN_WIDTH * K) * widths
    cumwidths = F.pad(torch.cumsum(widths, dim=-1), pad=(1, 0), value=0.0)
    cumwidths = (right - left)[..., None] * cumwidths + left[..., None]
    cumwidths[..., 0] = left
    cumwidths[..., -1] = right
    widths = cumwidths[..., 1:] - cumwidths[..., :-1]
    derivatives = MIN_DERIVATIVE + F.softplus(ud)
    heights = F.softmax(uh, dim=-1)
    heights = MIN_BIN_HEIGHT + (1 - MIN_BIN_HEIGHT * K) * heights
    cumheights = F.pad(torch.cumsum(heights, dim=-1), pad=(1, 0), value=0.0)
    cumheights = (top - bottom)[..., None] * cumheights + bottom[..., None]
    cumheights[..., 0] = bottom
    cumheights[..., -1] = top
    heights = cumheights[..., 1:] - cumheights[..., :-1]
    bin_idx = _searchsorted(cumheights if inverse else cumwidths, inputs)[..., None].clamp(0, K - 1)
    g = lambda t: t.gather(-1, bin_idx)[..., 0]          # noqa: E731
    in_cw, in_w, in_ch, in_h = g(cumwidths), g(widths), g(cumheights), g(heights)
    delta = heights / widths
    in_delta, in_d, in_d1 = g(delta), g(derivatives), g(derivatives[..., 1:])
    if inverse:
        a = (inputs - in_ch) * (in_d + in_d1 - 2 * in_delta) + in_h * (in_delta - in_d)
        b = in_h * in_d - (inputs - in_ch) * (in_d + in_d1 - 2 * in_delta)
        c = -in_delta * (inputs - in_ch)
        disc = b.pow(2) - 4 * a * c
        root = (2 * c) / (-b - torch.sqrt(disc))
        outputs = root * in_w + in_cw
        t1mt = root * (1 - root)
        denom = in_delta + (in_d + in_d1 - 2 * in_delta) * t1mt
        dnum = in_delta.pow(2) * (in_d1 * root.pow(2) + 2 * in_delta * t1mt + in_d * (1 - root).pow(2))
        return outputs, -(torch.log(dnum) - 2 * torch.log(denom))
    theta = (inputs - in_cw) / in_w
    t1mt = theta * (1 - theta)
    numer = in_h * (in_delta * theta.pow(2) + in_d * t1mt)
    denom = in_delta + (in_d + in_d1 - 2 * in_delta) * t1mt
    outputs = in_ch + numer / denom
    dnum = in_delta.pow(2) * (in_d1 * theta.pow(2) + 2 * in_delta * t1mt + in_d * (1 - theta).pow(2))
    return outputs, torch.log(dnum) - 2 * torch.log(denom)


def unconstrained_rqs(inputs, uw, uh, ud, inverse, circ: torch.Tensor, tail_bound: torch.Tensor):
    """Per-feature tails: `circ` [F] bool (circular) else linear, `tail_bound` [F].  inputs [B, F], uw/uh [B, F, K],
    ud [B, F, K + 1].  Linear tails: identity outside [-B, B], boundary derivatives fixed to 1; circular: last knot
    derivative tied to the first."""
    const = math.log(math.exp(1 - MIN_DERIVATIVE) - 1)
    ud = ud.clone()
    lin = ~circ
    ud[..., lin, 0] = const
    ud[..., lin, -1] = const
    ud[..., circ, -1] = ud[..., circ, 0]
    tb = torch.broadcast_to(tail_bound, inputs.shape)
    inside = (inputs >= -tb) & (inputs <= tb)
    outputs = inputs.clone()
    logabsdet = torch.zeros_like(inputs)
    if inside.any():
        o, l = rational_quadratic_spline(inputs[inside], uw[inside], uh[inside], ud[inside], inverse,
                                         -tb[inside], tb[inside], -tb[inside], tb[inside])
        outputs = outputs.masked_scatter(inside, o)
        logabsdet = logabsdet.masked_scatter(inside, l)
    return outputs, logabsdet


class PeriodicFeaturesElementwise(nn.Module):
    """x_i -> w_i0 sin(s_i x_i) + w_i1 cos(s_i x_i) on the listed features (weights initialised to one)."""

    def __init__(self, ndim: int, ind: Sequence[int], scale):
        super().__init__()
        self.register_buffer("ind", torch.as_tensor(list(ind), dtype=torch.long))
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.get_default_dtype()).reshape(-1))
        self.weights = nn.Parameter(torch.ones(len(ind), 2))

    def forward(self, x):
        if len(self.ind) == 0:
            return x
        xi = x[..., self.ind]
        xi = self.weights[:, 0] * torch.sin(self.scale * xi) + self.weights[:, 1] * torch.cos(self.scale * xi)
        out = x.clone()
        out[..., self.ind] = xi
        return out


class ResidualBlock(nn.Module):
    def __init__(self, features: int):
        super().__init__()
        self.linear_layers = nn.ModuleList([nn.Linear(features, features), nn.Linear(features, features)])
        nn.init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
        nn.init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def forward(self, x):
        t = self.linear_layers[0](F.relu(x))
        t = self.linear_layers[1](F.relu(t))
        return x + t


class ResidualNet(nn.Module):
    def __init__(self, in_features, out_features, hidden_features, num_blocks=1, preprocessing=None):
        super().__init__()
        self.hidden_features = hidden_features
        self.preprocessing = preprocessing
        self.initial_layer = nn.Linear(in_features, hidden_features)
        self.blocks = nn.ModuleList([ResidualBlock(hidden_features) for _ in range(num_blocks)])
        self.final_layer = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        t = x if self.preprocessing is None else self.preprocessing(x)
        t = self.initial_layer(t)
        for b in self.blocks:
            t = b(t)
        return self.final_layer(t)


class UnconditionalRQS(nn.Module):
    """Element-wise spline with its own parameters (identity at initialisation)."""

    def __init__(self, n_features, num_bins, circ, tail_bound):
        super().__init__()
        self.register_buffer("circ", circ)
        self.register_buffer("tail_bound", tail_bound)
        const = math.log(math.exp(1 - MIN_DERIVATIVE) - 1)
        self.unnormalized_widths = nn.Parameter(torch.zeros(n_features, num_bins))
        self.unnormalized_heights = nn.Parameter(torch.zeros(n_features, num_bins))
        self.unnormalized_derivatives = nn.Parameter(const * torch.ones(n_features, num_bins + 1))

    def forward(self, x, inverse=False):
        B = x.shape[0]
        e = lambda p: p[None].expand(B, *p.shape)          # noqa: E731
        return unconstrained_rqs(x, e(self.unnormalized_widths), e(self.unnormalized_heights),
                                 e(self.unnormalized_derivatives), inverse, self.circ, self.tail_bound)


class _PRQCT(nn.Module):
    """PiecewiseRationalQuadraticCouplingTransform with per-feature tails."""

    def __init__(self, mask, hidden, num_blocks, num_bins, circ_all, tail_bound_all, init_identity=True):
        super().__init__()
        D = mask.shape[0]
        feats = torch.arange(D)
        self.register_buffer("identity_features", feats[mask <= 0])
        self.register_buffer("transform_features", feats[mask > 0])
        self.num_bins = num_bins
        idf, trf = self.identity_features, self.transform_features
        circ_id = [i for i, f in enumerate(idf.tolist()) if bool(circ_all[f])]
        scale_pf = (math.pi / tail_bound_all[idf][circ_id]) if circ_id else torch.zeros(0)
        pf = PeriodicFeaturesElementwise(len(idf), circ_id, scale_pf) if circ_id else None
        self.transform_net = ResidualNet(len(idf), len(trf) * (3 * num_bins + 1), hidden, num_blocks, pf)
        if init_identity:
            nn.init.constant_(self.transform_net.final_layer.weight, 0.0)
            nn.init.constant_(self.transform_net.final_layer.bias, math.log(math.exp(1 - MIN_DERIVATIVE) - 1))
        self.register_buffer("circ_t", circ_all[trf].clone())
        self.register_buffer("tb_t", tail_bound_all[trf].clone())
        self.unconditional_transform = UnconditionalRQS(len(idf), num_bins, circ_all[idf].clone(), tail_bound_all[idf].clone())

    def _params(self, x_id):
        K = self.num_bins
        p = self.transform_net(x_id).view(x_id.shape[0], len(self.transform_features), 3 * K + 1)
        s = math.sqrt(self.transform_net.hidden_features)
        return p[..., :K] / s, p[..., K:2 * K] / s, p[..., 2 * K:]

    def forward(self, x):
        """spline EVALUATION direction (normflows wrapper: flow.inverse, i.e. the log_prob direction)."""
        x_id, x_tr = x[:, self.identity_features], x[:, self.transform_features]
        uw, uh, ud = self._params(x_id)
        y_tr, ld = unconstrained_rqs(x_tr, uw, uh, ud, False, self.circ_t, self.tb_t)
        y_id, ld_id = self.unconditional_transform(x_id, inverse=False)
        out = torch.empty_like(x)
        out[:, self.identity_features] = y_id
        out[:, self.transform_features] = y_tr
        return out, ld.sum(-1) + ld_id.sum(-1)

    def inverse(self, y):
        y_id, y_tr = y[:, self.identity_features], y[:, self.transform_features]
        x_id, ld_id = self.unconditional_transform(y_id, inverse=True)
        uw, uh, ud = self._params(x_id)
        x_tr, ld = unconstrained_rqs(y_tr, uw, uh, ud, True, self.circ_t, self.tb_t)
        out = torch.empty_like(y)
        out[:, self.identity_features] = x_id
        out[:, self.transform_features] = x_tr
        return out, ld.sum(-1) + ld_id.sum(-1)


class CircularCoupledRationalQuadraticSpline(nn.Module):
    def __init__(self, dim, num_blocks, hidden, ind_circ, tail_bound, num_bins, mask, init_identity=True):
        super().__init__()
        circ = torch.zeros(dim, dtype=torch.bool)
        circ[list(ind_circ)] = True
        self.prqct = _PRQCT(mask, hidden, num_blocks, num_bins, circ, tail_bound, init_identity)

    def forward(self, z):          # sampling direction: inverts the spline
        return self.prqct.inverse(z)

    def inverse(self, z):          # log_prob direction: evaluates the spline
        return self.prqct(z)


class PeriodicShift(nn.Module):
    def __init__(self, ind, bound, shift):
        super().__init__()
        self.register_buffer("ind", torch.as_tensor(list(ind), dtype=torch.long))
        self.register_buffer("bound", torch.as_tensor(bound, dtype=torch.get_default_dtype()).reshape(-1))
        self.register_buffer("shift", torch.as_tensor(shift, dtype=torch.get_default_dtype()).reshape(-1))

    def forward(self, z):
        z = z.clone()
        z[..., self.ind] = torch.remainder(z[..., self.ind] + self.shift + self.bound, 2 * self.bound) - self.bound
        return z, torch.zeros(len(z), dtype=z.dtype)

    def inverse(self, z):
        z = z.clone()
        z[..., self.ind] = torch.remainder(z[..., self.ind] - self.shift + self.bound, 2 * self.bound) - self.bound
        return z, torch.zeros(len(z), dtype=z.dtype)


class PeriodicWrap(nn.Module):
    def __init__(self, ind, bound):
        super().__init__()
        self.register_buffer("ind", torch.as_tensor(list(ind), dtype=torch.long))
        self.register_buffer("bound", torch.as_tensor(bound, dtype=torch.get_default_dtype()).reshape(-1))

    def forward(self, z):
        return z, torch.zeros(len(z), dtype=z.dtype)

    def inverse(self, z):
        z = z.clone()
        z[..., self.ind] = torch.remainder(z[..., self.ind] + self.bound, 2 * self.bound) - self.bound
        return z, torch.zeros(len(z), dtype=z.dtype)


class UniformGaussian(nn.Module):
    """Uniform on [-scale/2, scale/2] for the circular features, N(0, scale^2) for the others."""

    def __init__(self, ndim, ind_circ, scale):
        super().__init__()
        self.shape = (ndim,)
        circ = torch.zeros(ndim, dtype=torch.bool)
        circ[list(ind_circ)] = True
        self.register_buffer("circ", circ)
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.get_default_dtype()).reshape(-1))

    def forward_eps(self, u, eps):
        """u [B, D] ~ U(0, 1), eps [B, D] ~ N(0, 1): the circular features use u - 0.5, the others eps."""
        z = torch.where(self.circ, u - 0.5, eps) * self.scale
        return z, self.log_prob(z)

    def log_prob(self, z):
        lu = -torch.log(self.scale)
        lg = -0.5 * math.log(2 * math.pi) - torch.log(self.scale) - 0.5 * (z / self.scale) ** 2
        return torch.where(self.circ, lu.expand_as(z), lg).sum(-1)


class SplineFlow(nn.Module):
    """NormalizingFlow(q0, flows): sample = forwards in order (log q -= log det), log_prob = inverses in reverse."""

    def __init__(self, q0, flows: List[nn.Module]):
        super().__init__()
        self.q0 = q0
        self.flows = nn.ModuleList(flows)

    def sample_eps(self, u, eps):
        z, log_q = self.q0.forward_eps(u, eps)
        for f in self.flows:
            z, ld = f(z)
            log_q = log_q - ld
        return z, log_q

    def log_prob(self, x):
        log_q = torch.zeros(len(x), dtype=x.dtype)
        z = x
        for f in reversed(self.flows):
            z, ld = f.inverse(z)
            log_q = log_q + ld
        return log_q + self.q0.log_prob(z)


def make_circular_coupled_flow(dim: int, n_layers: int, hidden: int, ind_circ: Sequence[int], tail_bound, num_bins: int = 8,
                               num_blocks: int = 1, seed: int = 0, circ_shift: str = "random",
                               init_identity: bool = True) -> SplineFlow:
    """The layer list of make_aldp_model.py:121-134,146-167 ('circular-coup-nsf', mixing null, actnorm false)."""
    tail_bound = torch.as_tensor(tail_bound, dtype=torch.get_default_dtype()).reshape(-1)
    if tail_bound.numel() == 1:
        tail_bound = tail_bound.expand(dim).clone()
    ind_circ = list(ind_circ)
    bound_circ = tail_bound[ind_circ]
    scale = torch.ones(dim)
    scale[ind_circ] = 2 * bound_circ
    layers, mask = [], None
    for i in range(n_layers):
        if i % 2 == 0:                     # nf.utils.masks.create_random_binary_mask(ndim, seed=seed + i)
            g = torch.Generator().manual_seed(seed + i)
            mask = torch.zeros(dim)
            weights = torch.ones(dim)
            num_samples = dim // 2 + dim % 2
            mask[torch.multinomial(weights, num_samples, replacement=False, generator=g)] += 1
        else:
            mask = 1 - mask
        layers.append(CircularCoupledRationalQuadraticSpline(dim, num_blocks, hidden, ind_circ, tail_bound, num_bins,
                                                             mask.clone(), init_identity))
        if i % 2 == 1 and i != n_layers - 1 and circ_shift is not None and len(ind_circ):
            if circ_shift == "constant":
                layers.append(PeriodicShift(ind_circ, bound_circ, bound_circ))
            else:
                g = torch.Generator().manual_seed(seed + i)
                layers.append(PeriodicShift(ind_circ, bound_circ, (torch.rand([], generator=g) + 0.5) * bound_circ))
    if len(ind_circ):
        layers.append(PeriodicWrap(ind_circ, bound_circ))
    return SplineFlow(UniformGaussian(dim, ind_circ, scale), layers)


def randomize(flow: SplineFlow, std: float = 0.3, seed: int = 0) -> None:
    """Non-trivial parameters for parity tests (an identity-initialised flow would hide layout errors)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for f in flow.flows:
            if isinstance(f, CircularCoupledRationalQuadraticSpline):
                net = f.prqct.transform_net
                net.final_layer.weight.copy_(torch.randn(net.final_layer.weight.shape, generator=g) * std / 4)
                net.final_layer.bias.add_(torch.randn(net.final_layer.bias.shape, generator=g) * std)
                lin = net.blocks[0].linear_layers[1]
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.05)
                u = f.prqct.unconditional_transform
                u.unnormalized_widths.add_(torch.randn(u.unnormalized_widths.shape, generator=g) * std)
                u.unnormalized_heights.add_(torch.randn(u.unnormalized_heights.shape, generator=g) * std)
                u.unnormalized_derivatives.add_(torch.randn(u.unnormalized_derivatives.shape, generator=g) * std)
                if net.preprocessing is not None:
                    net.preprocessing.weights.add_(torch.randn(net.preprocessing.weights.shape, generator=g) * 0.2)
