"""RealNVP `TrainableDistribution` backed by the HIP kernels.

Host-side mirror of what the reference builds with normflows in
experiments/make_flow/make_normflow_model.py:11-30,82-96 (`make_wrapped_normflow_realnvp`) and wraps
in fab/wrappers/normflows.py:8-31 (`WrappedNormFlowModel`): same constructor meaning, same
`Distribution` methods (fab/types_.py:8-27), same state-dict key names
(`_nf_model.q0.loc`, `_nf_model.flows.{2i}.flows.1.param_map.net.{0,2,4}.{weight,bias}`,
`_nf_model.flows.{2i+1}.{P,L,U,log_S,sign_S,eye}`) so reference checkpoints load (fab/core.py:237-240).

Hot path (no autograd graph requested): `sample_and_log_prob`, `log_prob`, `log_prob_and_grad` run the
fp32-MFMA kernels of csrc/flow_kernels.hip through the C ABI.  When autograd is recording w.r.t. the
parameters (the trainer's `flow.log_prob(x)` + `loss.backward()`, fab/train_with_prioritised_buffer.py:162-173)
`log_prob` runs the HIP forward with a tape and `backward` the parameter-gradient GEMM kernels of
csrc/train_kernels.hip (`_LogProbWithTape`).  There is no CPU path and no stock-PyTorch density path: every entry
raises `FabhipError` for tensors that are not on the GPU.  The one exception is documented at `_aten_sample`:
the REPARAMETERISED sampling gradient needed only by the non-FAB baseline losses (`flow_reverse_kl`,
`flow_alpha_2_div_nis`, fab/core.py:130-152) is expressed with ATen ops on the GPU.
"""
import ctypes as C
import math
from typing import Tuple

import torch
import torch.nn as nn

from . import _lib


class _MLP(nn.Module):
    def __init__(self, layers, init_zeros=True):
        super().__init__()
        net = []
        for k in range(len(layers) - 2):
            net += [nn.Linear(layers[k], layers[k + 1]), nn.LeakyReLU(0.0)]
        net.append(nn.Linear(layers[-2], layers[-1]))
        if init_zeros:
            nn.init.zeros_(net[-1].weight)
            nn.init.zeros_(net[-1].bias)
        self.net = nn.Sequential(*net)


class _Stub(nn.Module):
    """parameter-less placeholder keeping normflows' module indices (Split / Merge)."""


class _AffineCoupling(nn.Module):
    def __init__(self, param_map):
        super().__init__()
        self.param_map = param_map


class _AffineCouplingBlock(nn.Module):
    def __init__(self, param_map):
        super().__init__()
        self.flows = nn.ModuleList([_Stub(), _AffineCoupling(param_map), _Stub()])


class _InvertibleAffine(nn.Module):
    def __init__(self, dim):
        super().__init__()
        Q, _ = torch.linalg.qr(torch.randn(dim, dim))
        P, L, U = torch.linalg.lu(Q)
        S = U.diag()
        self.register_buffer("P", P)
        self.L = nn.Parameter(L)
        self.register_buffer("sign_S", torch.sign(S))
        self.log_S = nn.Parameter(torch.log(torch.abs(S)))
        self.U = nn.Parameter(torch.triu(U, diagonal=1))
        self.register_buffer("eye", torch.diag(torch.ones(dim)))

    def assemble(self, inverse=False):
        L = torch.tril(self.L, diagonal=-1) + self.eye
        U = torch.triu(self.U, diagonal=1) + torch.diag(self.sign_S * torch.exp(self.log_S))
        if inverse:
            L_inv = torch.inverse(L.double()).type(self.log_S.dtype)
            U_inv = torch.inverse(U.double()).type(self.log_S.dtype)
            return U_inv @ L_inv @ self.P.t()
        return self.P @ L @ U


class _DiagGaussian(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.shape = (dim,)
        self.loc = nn.Parameter(torch.zeros(1, dim))
        self.log_scale = nn.Parameter(torch.zeros(1, dim))


class _NormalizingFlow(nn.Module):
    def __init__(self, dim, n_layers, width):
        super().__init__()
        self.q0 = _DiagGaussian(dim)
        d = int((dim / 2) + 0.5)
        flows = []
        for _ in range(n_layers):
            flows.append(_AffineCouplingBlock(_MLP([d, width, width, 2 * (dim - d)], init_zeros=True)))
            flows.append(_InvertibleAffine(dim))
        self.flows = nn.ModuleList(flows)


class _LogProbWithTape(torch.autograd.Function):
    """log q(x) through fabhip_flow_log_prob_tape; backward = fabhip_flow_param_grad (one flat gradient image,
    handed to autograd as views) and grad_output * d log q / dx.  Replaces the autograd graph the reference
    builds for `flow.log_prob(x)` (fab/train_with_prioritised_buffer.py:162-173, fab/core.py:112-118)."""

    @staticmethod
    def forward(ctx, flow, x, *params):
        if x.requires_grad:
            log_q, handle, grad_x = flow.log_prob_with_tape(x, want_grad_x=True)
        else:
            (log_q, handle), grad_x = flow.log_prob_with_tape(x), None
        ctx.flow, ctx.handle = flow, handle[1:]
        ctx.save_for_backward(handle[0], grad_x if grad_x is not None else torch.empty(0, device=log_q.device))
        return log_q

    @staticmethod
    def backward(ctx, g):
        flow = ctx.flow
        tape, grad_x = ctx.saved_tensors
        coef = g.detach().contiguous().float()
        flat = flow.param_grad_flat((tape,) + ctx.handle, coef)
        flow._last_flat_grad = flat
        gx = coef[:, None] * grad_x if ctx.needs_input_grad[1] else None
        if len(ctx.needs_input_grad) == 3:                   # flat mode: one leaf holds every parameter (FlatAdam)
            return None, gx, flat
        views = flow._grad_views(flat)
        grads = [v if need else None for v, need in zip(views, ctx.needs_input_grad[2:])]
        return (None, gx, *grads)


class RealNVP(nn.Module):
    """`make_wrapped_normflow_realnvp(dim, n_flow_layers, layer_nodes_per_dim, act_norm=False)`."""

    def __init__(self, dim: int, n_flow_layers: int = 5, layer_nodes_per_dim: int = 10, act_norm: bool = False):
        super().__init__()
        if act_norm:
            raise NotImplementedError("ActNorm layers are not part of the MI355X hot path (all shipped "
                                      "reference configs set act_norm=false)")
        self.dim, self.n_layers, self.width = dim, n_flow_layers, dim * layer_nodes_per_dim
        self.d = int((dim / 2) + 0.5)
        self._nf_model = _NormalizingFlow(dim, n_flow_layers, self.width)
        self._packed = None
        self._packed_key = None
        self._params_struct = None
        self._grad_layout = None
        self._flat_leaf = None
        self._packed_has_inverse = False

    # ---- Distribution interface (fab/types_.py:8-27) ---------------------------------------------
    @property
    def event_shape(self) -> Tuple[int, ...]:
        return self._nf_model.q0.shape

    def sample_and_log_prob(self, shape: Tuple[int, ...], eps: torch.Tensor = None):
        assert len(shape) == 1
        dev = self._nf_model.q0.loc.device
        if eps is None:
            eps = torch.randn((shape[0], self.dim), dtype=torch.float32, device=dev)
        if torch.is_grad_enabled() and self._params_need_grad():
            return self._aten_sample(eps)
        return self.native_sample(eps)

    def sample(self, shape: Tuple) -> torch.Tensor:
        return self.sample_and_log_prob(shape)[0]

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x, "x")
        if torch.is_grad_enabled() and (x.requires_grad or self._params_need_grad()):
            if self._flat_leaf is not None:                  # parameters live in one buffer (optim.FlatAdam)
                return _LogProbWithTape.apply(self, x, self._flat_leaf)
            return _LogProbWithTape.apply(self, x, *self._grad_tensors())
        return self.native_log_prob(x)[0]

    # ---- training path: HIP forward with a tape + parameter-gradient GEMMs (csrc/train_kernels.hip) -----

    def _grad_tensors(self):
        """Parameters in the order of the flat gradient image (fabhip_flow_grad_layout)."""
        out = []
        for l1, l2, l3, aff in self._layers():
            out += [l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, aff.L, aff.U, aff.log_S]
        q0 = self._nf_model.q0
        return out + [q0.loc, q0.log_scale]

    def _grad_views(self, flat: torch.Tensor):
        """Views of the flat gradient image, one per tensor of `_grad_tensors()`."""
        if self._grad_layout is None:
            lay = (C.c_int64 * 13)()
            _lib.check(_lib.load().fabhip_flow_grad_layout(self.dim, self.n_layers, self.width, lay), "grad_layout")
            self._grad_layout = [int(v) for v in lay]
        stride, w1, b1, w2, b2, w3, b3, oL, oU, oS, loc, lsc, _ = self._grad_layout
        D, d, W = self.dim, self.d, self.width
        shapes = [(w1, (W, d)), (b1, (W,)), (w2, (W, W)), (b2, (W,)), (w3, (2 * (D - d), W)), (b3, (2 * (D - d),)),
                  (oL, (D, D)), (oU, (D, D)), (oS, (D,))]
        views = []
        for k in range(self.n_layers):
            base = k * stride
            for off, shp in shapes:
                n = 1
                for s in shp:
                    n *= s
                views.append(flat[base + off: base + off + n].view(shp))
        views.append(flat[loc: loc + D].view(1, D))
        views.append(flat[lsc: lsc + D].view(1, D))
        return views

    # ---- native (HIP) entry points ------------------------------------------------------------------
    def _params_need_grad(self):
        return any(p.requires_grad for p in self.parameters())

    def _layers(self):
        fl = self._nf_model.flows
        for i in range(self.n_layers):
            net = fl[2 * i].flows[1].param_map.net
            yield net[0], net[2], net[4], fl[2 * i + 1]

    def native(self, need_inverse: bool = True):
        """(Flow struct, packed image) — re-tiled by the pack kernels whenever a parameter changed.
        need_inverse=False (density evaluations only, e.g. the minibatch loop of the trainer) skips the W^-1
        matrices; the next caller that samples gets a full re-pack."""
        lib = _lib.load()
        q0 = self._nf_model.q0
        _lib.require_device(q0.loc, "RealNVP parameters")
        tensors = []
        for l1, l2, l3, aff in self._layers():
            tensors += [l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, aff.L, aff.U, aff.log_S,
                        aff.sign_S, aff.P]
        tensors += [q0.loc, q0.log_scale]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._packed_key or (need_inverse and not self._packed_has_inverse):
            for t in tensors:
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise _lib.FabhipError("RealNVP parameters must be contiguous float32 for the HIP path")
            n = lib.fabhip_flow_packed_floats(self.dim, self.n_layers, self.width)
            if n < 0:
                raise _lib.FabhipError(f"flow shape not supported by the kernels: dim={self.dim} width={self.width}")
            if self._packed is None or self._packed.numel() != n or self._packed.device != q0.loc.device:
                self._packed = torch.empty(n, dtype=torch.float32, device=q0.loc.device)
            p = _lib.FlowParams()
            p.dim, p.n_layers, p.width = self.dim, self.n_layers, self.width
            names = ("w1", "b1", "w2", "b2", "w3", "b3", "lu_L", "lu_U", "log_S", "sign_S", "perm_P")
            for k in range(self.n_layers):
                for j, nm in enumerate(names):
                    getattr(p, nm)[k] = tensors[11 * k + j].data_ptr()
            p.loc, p.log_scale = q0.loc.data_ptr(), q0.log_scale.data_ptr()
            pack = lib.fabhip_flow_pack if need_inverse else lib.fabhip_flow_pack_density
            _lib.check(pack(C.byref(p), _lib.ptr(self._packed), _lib.stream_ptr()), "flow_pack")
            self._packed_has_inverse = need_inverse
            self._packed_key = key
            self._params_struct = p
        f = _lib.Flow(self.dim, self.n_layers, self.width, self._packed.data_ptr())
        return f, self._packed

    def native_sample(self, eps: torch.Tensor):
        lib = _lib.load()
        _lib.require_device(eps, "eps")
        f, _ = self.native()
        eps = eps.contiguous().float()
        B = eps.shape[0]
        x = torch.empty_like(eps)
        log_q = torch.empty(B, dtype=torch.float32, device=eps.device)
        _lib.check(lib.fabhip_flow_sample(C.byref(f), _lib.ptr(eps), _lib.ptr(x), _lib.ptr(log_q), B,
                                          _lib.stream_ptr()), "flow_sample")
        return x, log_q

    def native_log_prob(self, x: torch.Tensor, with_grad: bool = False):
        lib = _lib.load()
        _lib.require_device(x, "x")
        f, _ = self.native(need_inverse=False)
        x = x.detach().contiguous().float()
        B = x.shape[0]
        log_q = torch.empty(B, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if with_grad else None
        _lib.check(lib.fabhip_flow_log_prob(C.byref(f), _lib.ptr(x), _lib.ptr(log_q), _lib.ptr(grad), B,
                                            _lib.stream_ptr()), "flow_log_prob")
        return log_q, grad

    # ---- autograd-free training entry points (what `_LogProbWithTape` wraps) -----------------------------------------
    def log_prob_with_tape(self, x: torch.Tensor, want_grad_x: bool = False):
        """(log q(x), tape handle[, d log q / dx]) through fabhip_flow_log_prob_tape, no autograd graph."""
        lib = _lib.load()
        _lib.require_device(x, "x")
        f, _ = self.native(need_inverse=False)
        xd = x.detach().contiguous().float()
        B = xd.shape[0]
        log_q = torch.empty(B, dtype=torch.float32, device=xd.device)
        grad_x = torch.empty_like(xd) if want_grad_x else None
        nbytes = lib.fabhip_flow_tape_bytes(self.dim, self.n_layers, self.width, B)
        tape = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=xd.device)
        _lib.check(lib.fabhip_flow_log_prob_tape(C.byref(f), _lib.ptr(xd), _lib.ptr(log_q), _lib.ptr(grad_x), B,
                                                 _lib.ptr(tape), nbytes, _lib.stream_ptr()), "flow_log_prob_tape")
        handle = (tape, nbytes, B, self._packed_key)
        return (log_q, handle, grad_x) if want_grad_x else (log_q, handle)

    def param_grad_flat(self, tape_handle, coef: torch.Tensor) -> torch.Tensor:
        """sum_b coef[b] * d log q(x_b) / d theta as one flat gradient image (layout: `_grad_views`)."""
        lib = _lib.load()
        tape, nbytes, B, key = tape_handle
        if self._packed_key != key:
            raise _lib.FabhipError("flow parameters were modified between log_prob_with_tape(x) and param_grad_flat()")
        f, _ = self.native(need_inverse=False)
        c = coef.detach().contiguous().float()
        flat = torch.empty(lib.fabhip_flow_grad_floats(self.dim, self.n_layers, self.width), dtype=torch.float32,
                           device=c.device)
        _lib.check(lib.fabhip_flow_param_grad(C.byref(self._params_struct), C.byref(f), _lib.ptr(tape), nbytes,
                                              _lib.ptr(c), B, _lib.ptr(flat), _lib.stream_ptr()), "flow_param_grad")
        return flat

    def log_prob_and_grad(self, x: torch.Tensor):
        """(log q(x), d log q / dx) — what `grad_and_value(x, flow.log_prob)` computes (base.py:50-56)."""
        return self.native_log_prob(x, with_grad=True)

    # ---- reparameterised sampling gradient (baseline losses only, NOT on the FAB path) ------------------------------
    def _aten_sample(self, eps):
        """x, log q = flow.sample with an autograd graph w.r.t. the parameters, for `flow_reverse_kl` /
        `flow_alpha_2_div_nis` (fab/core.py:130-152), the paper's non-FAB baselines.  GPU only (no CPU path)."""
        _lib.require_device(eps, "eps")
        q0 = self._nf_model.q0
        z = q0.loc + torch.exp(q0.log_scale) * eps
        log_q = -0.5 * self.dim * math.log(2 * math.pi) - torch.sum(q0.log_scale + 0.5 * torch.pow(eps, 2), 1)
        relu = torch.nn.functional.relu
        for l1, l2, l3, aff in self._layers():
            z1, z2 = z[:, :self.d], z[:, self.d:]
            prm = l3(relu(l2(relu(l1(z1)))))
            shift, scale = prm[:, 0::2], prm[:, 1::2]
            z2 = z2 * torch.exp(scale) + shift
            log_q = log_q - torch.sum(scale, dim=1)
            z = torch.cat([z1, z2], 1) @ aff.assemble(inverse=True)
            log_q = log_q + torch.sum(aff.log_S)
        return z, log_q


def make_wrapped_normflow_realnvp(dim: int, n_flow_layers: int = 5, layer_nodes_per_dim: int = 10,
                                  act_norm: bool = True) -> RealNVP:
    """Same name/arguments as experiments/make_flow/make_normflow_model.py:82-96."""
    return RealNVP(dim, n_flow_layers, layer_nodes_per_dim, act_norm)
