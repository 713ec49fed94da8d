"""Oracle RealNVP flow (PyTorch CPU) — TEST INFRASTRUCTURE, never imported by the product.

Restates the `normflows` RealNVP that the reference builds in
experiments/make_flow/make_normflow_model.py:11-30 (layer list) and :82-96
(`make_wrapped_normflow_realnvp`) and wraps in fab/wrappers/normflows.py:8-31.

`normflows` itself is a third-party dependency (requirements.txt:3, unpinned) that is
absent from /root/reference and cannot be installed offline -> the arithmetic below is
restated from the library's published definition (RealNVP affine coupling with
`scale_map="exp"`, channel split, LU-parametrised invertible affine, diagonal Gaussian
base).  **Parity unpinned** by the reference: its only tests at this boundary assert
shapes (fab/wrappers/normflow_test.py:33-34, fab/sampling_methods/base_test.py:12-24).

Module / parameter names follow normflows so that a real normflows ``state_dict`` has
the same keys:  ``q0.loc``, ``q0.log_scale``, ``flows.{2i}.flows.1.param_map.net.{0,2,4}.
{weight,bias}``, ``flows.{2i+1}.{P,L,U,log_S,sign_S,eye}``.
"""
from typing import List, Tuple

import math
import torch
import torch.nn as nn


class MLP(nn.Module):
    """Linear -> LeakyReLU(leaky) -> ... -> Linear  (normflows nets.MLP; leaky=0.0 default)."""

    def __init__(self, layers: List[int], leaky: float = 0.0, init_zeros: bool = False):
        super().__init__()
        net = []
        for k in range(len(layers) - 2):
            net.append(nn.Linear(layers[k], layers[k + 1]))
            net.append(nn.LeakyReLU(leaky))
        net.append(nn.Linear(layers[-2], layers[-1]))
        if init_zeros:
            nn.init.zeros_(net[-1].weight)
            nn.init.zeros_(net[-1].bias)
        self.net = nn.Sequential(*net)

    def forward(self, x):
        return self.net(x)


class Split(nn.Module):
    """Channel split: z -> (z[:, :ceil(D/2)], z[:, ceil(D/2):])  (torch.chunk semantics)."""

    def forward(self, z):
        z1, z2 = z.chunk(2, dim=1)
        return [z1, z2], 0

    def inverse(self, z):
        z1, z2 = z
        return torch.cat([z1, z2], 1), 0


class Merge(Split):
    def forward(self, z):
        return super().inverse(z)

    def inverse(self, z):
        return super().forward(z)


class AffineCoupling(nn.Module):
    """z2' = z2 * exp(s(z1)) + t(z1);  params interleaved: t = h[:, 0::2], s = h[:, 1::2]."""

    def __init__(self, param_map: nn.Module):
        super().__init__()
        self.add_module("param_map", param_map)

    def forward(self, z):
        z1, z2 = z
        param = self.param_map(z1)
        shift = param[:, 0::2]
        scale_ = param[:, 1::2]
        z2 = z2 * torch.exp(scale_) + shift
        log_det = torch.sum(scale_, dim=1)
        return [z1, z2], log_det

    def inverse(self, z):
        z1, z2 = z
        param = self.param_map(z1)
        shift = param[:, 0::2]
        scale_ = param[:, 1::2]
        z2 = (z2 - shift) * torch.exp(-scale_)
        log_det = -torch.sum(scale_, dim=1)
        return [z1, z2], log_det


class AffineCouplingBlock(nn.Module):
    """Split -> AffineCoupling -> Merge."""

    def __init__(self, param_map: nn.Module):
        super().__init__()
        self.flows = nn.ModuleList([Split(), AffineCoupling(param_map), Merge()])

    def forward(self, z):
        log_det_tot = 0
        for f in self.flows:
            z, ld = f(z)
            log_det_tot = log_det_tot + ld
        return z, log_det_tot

    def inverse(self, z):
        log_det_tot = 0
        for f in reversed(self.flows):
            z, ld = f.inverse(z)
            log_det_tot = log_det_tot + ld
        return z, log_det_tot


class InvertibleAffine(nn.Module):
    """z -> z @ W^-1 (forward) / z @ W (inverse), W = P (tril(L,-1)+I) (triu(U,1)+diag(sign_S e^log_S)).

    Initialised from the QR of a random Gaussian matrix; L^-1, U^-1 are taken in float64
    and cast back (as normflows does for non-double parameters)."""

    def __init__(self, num_channels: int):
        super().__init__()
        self.num_channels = num_channels
        Q, _ = torch.linalg.qr(torch.randn(num_channels, num_channels))
        P, L, U = torch.linalg.lu(Q)
        self.register_buffer("P", P)
        self.L = nn.Parameter(L)
        S = U.diag()
        self.register_buffer("sign_S", torch.sign(S))
        self.log_S = nn.Parameter(torch.log(torch.abs(S)))
        self.U = nn.Parameter(torch.triu(U, diagonal=1))
        self.register_buffer("eye", torch.diag(torch.ones(num_channels)))

    def _assemble_W(self, inverse: bool = False):
        L = torch.tril(self.L, diagonal=-1) + self.eye
        U = torch.triu(self.U, diagonal=1) + torch.diag(self.sign_S * torch.exp(self.log_S))
        if inverse:
            if self.log_S.dtype == torch.float64:
                L_inv = torch.inverse(L)
                U_inv = torch.inverse(U)
            else:
                L_inv = torch.inverse(L.double()).type(self.log_S.dtype)
                U_inv = torch.inverse(U.double()).type(self.log_S.dtype)
            W = U_inv @ L_inv @ self.P.t()
        else:
            W = self.P @ L @ U
        return W

    def forward(self, z):
        W = self._assemble_W(inverse=True)
        return z @ W, -torch.sum(self.log_S)

    def inverse(self, z):
        W = self._assemble_W()
        return z @ W, torch.sum(self.log_S)


class ActNorm(nn.Module):
    """normflows ActNorm (flows/normalization.py; AffineConstFlow of flows/affine/coupling.py with scale and shift):
    forward z * exp(s) + t with log_det sum(s), inverse (z - t) * exp(-s) with log_det -sum(s); the first batch
    through either direction initialises s, t from its statistics (torch.std: unbiased).  normflows is absent from the
    reference tree (requirements.txt:3, version unpinned): restated from its published definition, parity unpinned."""

    def __init__(self, dim: int):
        super().__init__()
        self.s = nn.Parameter(torch.zeros(1, dim))
        self.t = nn.Parameter(torch.zeros(1, dim))
        self.register_buffer("data_dep_init_done", torch.tensor(0.0))

    def forward(self, z):
        if not self.data_dep_init_done > 0.0:
            s_init = -torch.log(z.std(dim=0, keepdim=True) + 1e-6)
            self.s.data = s_init.data
            self.t.data = (-z.mean(dim=0, keepdim=True) * torch.exp(self.s)).data
            self.data_dep_init_done = torch.tensor(1.0)
        return z * torch.exp(self.s) + self.t, torch.sum(self.s)

    def inverse(self, z):
        if not self.data_dep_init_done > 0.0:
            s_init = torch.log(z.std(dim=0, keepdim=True) + 1e-6)
            self.s.data = s_init.data
            self.t.data = z.mean(dim=0, keepdim=True).data
            self.data_dep_init_done = torch.tensor(1.0)
        return (z - self.t) * torch.exp(-self.s), -torch.sum(self.s)


class DiagGaussian(nn.Module):
    def __init__(self, shape: int):
        super().__init__()
        self.shape = (shape,)
        self.d = shape
        self.loc = nn.Parameter(torch.zeros(1, shape))
        self.log_scale = nn.Parameter(torch.zeros(1, shape))

    def forward_eps(self, eps):
        """Sample with explicit standard-normal noise eps[B, D] -> (z, log_p)."""
        z = self.loc + torch.exp(self.log_scale) * eps
        log_p = -0.5 * self.d * math.log(2 * math.pi) - torch.sum(
            self.log_scale + 0.5 * torch.pow(eps, 2), 1)
        return z, log_p

    def forward(self, num_samples=1):
        eps = torch.randn((num_samples,) + self.shape, dtype=self.loc.dtype)
        return self.forward_eps(eps)

    def log_prob(self, z):
        return -0.5 * self.d * math.log(2 * math.pi) - torch.sum(
            self.log_scale + 0.5 * torch.pow((z - self.loc) / torch.exp(self.log_scale), 2), 1)


class NormalizingFlow(nn.Module):
    def __init__(self, q0: nn.Module, flows: List[nn.Module]):
        super().__init__()
        self.q0 = q0
        self.flows = nn.ModuleList(flows)

    def sample_eps(self, eps):
        z, log_q = self.q0.forward_eps(eps)
        for f in self.flows:
            z, ld = f(z)
            log_q = log_q - ld
        return z, log_q

    def sample(self, num_samples=1):
        eps = torch.randn((num_samples,) + self.q0.shape, dtype=self.q0.loc.dtype)
        return self.sample_eps(eps)

    def log_prob(self, x):
        log_q = torch.zeros(len(x), dtype=x.dtype)
        z = x
        for i in range(len(self.flows) - 1, -1, -1):
            z, ld = self.flows[i].inverse(z)
            log_q = log_q + ld
        log_q = log_q + self.q0.log_prob(z)
        return log_q


def make_realnvp(dim: int, n_flow_layers: int, layer_nodes_per_dim: int, act_norm: bool = False) -> NormalizingFlow:
    """experiments/make_flow/make_normflow_model.py:11-30."""
    flows = []
    width = dim * layer_nodes_per_dim
    for _ in range(n_flow_layers):
        d = int((dim / 2) + 0.5)
        param_map = MLP([d, width, width, 2 * (dim - d)], init_zeros=True)
        flows.append(AffineCouplingBlock(param_map))
        flows.append(InvertibleAffine(dim))
        if act_norm:
            flows.append(ActNorm(dim))
    return NormalizingFlow(DiagGaussian(dim), flows)


def randomize_last_layers(nf: NormalizingFlow, std: float = 0.05, seed: int = 1234) -> None:
    """The reference's ``init_zeros=True`` makes an untrained flow a pure linear map.  For
    non-trivial log-dets in benchmarks / fixtures re-draw the last coupling Linear N(0, std^2)
     and give the base a non-trivial loc / log_scale."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for f in nf.flows:
            if isinstance(f, AffineCouplingBlock):
                last = f.flows[1].param_map.net[-1]
                last.weight.copy_(torch.randn(last.weight.shape, generator=g) * std)
                last.bias.copy_(torch.randn(last.bias.shape, generator=g) * std)
        nf.q0.loc.copy_(torch.randn(nf.q0.loc.shape, generator=g) * 0.1)
        nf.q0.log_scale.copy_(torch.randn(nf.q0.log_scale.shape, generator=g) * 0.1)


class WrappedFlow(nn.Module):
    """fab/wrappers/normflows.py:8-31 — the `Distribution` plug-in over the oracle flow."""

    def __init__(self, nf: NormalizingFlow):
        super().__init__()
        self._nf_model = nf

    def sample_and_log_prob(self, shape: Tuple[int, ...]):
        assert len(shape) == 1
        return self._nf_model.sample(shape[0])

    def sample_and_log_prob_eps(self, eps):
        return self._nf_model.sample_eps(eps)

    def sample(self, shape):
        return self.sample_and_log_prob(shape)[0]

    def log_prob(self, x):
        return self._nf_model.log_prob(x)

    @property
    def event_shape(self):
        return self._nf_model.q0.shape
