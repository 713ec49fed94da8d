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
csrc/train_kernels.hip (the custom op `fabhip::realnvp_logprob_tape` and its registered autograd, _ops.py); the
REPARAMETERISED sampling gradient of the non-FAB baseline losses (`flow_reverse_kl`, `flow_alpha_2_div_nis`,
fab/core.py:130-152) is the custom op `fabhip::realnvp_sample_tape` (HIP sampler forward, `k_flow_sample_bwd` + the same
parameter-gradient kernels backward).  There is no CPU path and no stock-PyTorch path: every entry raises `FabhipError`
for tensors that are not on the GPU.
"""
import math
from typing import Tuple

import torch
import torch.nn as nn

from . import _ops


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


class _ActNorm(nn.Module):
    """normflows ActNorm(dim) (AffineConstFlow with data-dependent initialisation): sampling direction
    z <- z * exp(s) + t, log_det = sum(s); `s`, `t` of shape [1, dim] and the `data_dep_init_done` buffer under the
    normflows names."""

    def __init__(self, dim):
        super().__init__()
        self.s = nn.Parameter(torch.zeros(1, dim))
        self.t = nn.Parameter(torch.zeros(1, dim))
        self.register_buffer("data_dep_init_done", torch.tensor(0.0))


class _DiagGaussian(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.shape = (dim,)
        self.loc = nn.Parameter(torch.zeros(1, dim))
        self.log_scale = nn.Parameter(torch.zeros(1, dim))


class _NormalizingFlow(nn.Module):
    def __init__(self, dim, n_layers, width, act_norm=False):
        super().__init__()
        self.q0 = _DiagGaussian(dim)
        d = int((dim / 2) + 0.5)
        flows = []
        for _ in range(n_layers):
            flows.append(_AffineCouplingBlock(_MLP([d, width, width, 2 * (dim - d)], init_zeros=True)))
            flows.append(_InvertibleAffine(dim))
            if act_norm:                                   # make_normflow_model.py:27-29
                flows.append(_ActNorm(dim))
        self.flows = nn.ModuleList(flows)


class RealNVP(nn.Module):
    """`make_wrapped_normflow_realnvp(dim, n_flow_layers, layer_nodes_per_dim, act_norm)`.  With `act_norm` an
    ActNorm follows every InvertibleAffine (make_normflow_model.py:27-29); the kernels see it folded into the affine
    map (W' = diag(e^-s) W, W'^-1 = W^-1 diag(e^s), additive terms, log-det), its gradients come out of the
    LU chain-rule kernel.  Like the reference builder (make_normflow_model.py:94-95) call `init_act_norm()` (one
    500-sample draw per layer on the GPU) - or load a checkpoint - before use: `make_wrapped_normflow_realnvp` does."""

    def __init__(self, dim: int, n_flow_layers: int = 5, layer_nodes_per_dim: int = 10, act_norm: bool = False):
        super().__init__()
        self.dim, self.n_layers, self.width = dim, n_flow_layers, dim * layer_nodes_per_dim
        self.d = int((dim / 2) + 0.5)
        self.act_norm = bool(act_norm)
        self._act_norm_ready = False
        # None: follow the process default (`fab_torch_amd.fast_mode`); "fp32" / "fast": every density + gradient /
        # transition / AIS call that evaluates THIS flow runs the parity / the bf16 fast-mode kernels (per call:
        # fabhip_flow::precision), so two samplers of one process can differ
        self.precision = None
        self._nf_model = _NormalizingFlow(dim, n_flow_layers, self.width, self.act_norm)
        self._packed = None
        self._packed_key = None
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
        if torch.is_grad_enabled() and (eps.requires_grad or self._params_need_grad()):
            # the differentiable sampling op (reparameterised baseline losses, fab/core.py:130-152): forward = the HIP
            # sampler, backward (registered in _ops.py) = fabhip::realnvp_sample_grad_tape + fabhip::realnvp_param_grad
            _ops.require_device(eps, "eps")
            ops = _ops.load()
            packed, D, K, W = self.native()
            theta = self._flat_leaf if self._flat_leaf is not None else \
                torch.cat([p.reshape(-1) for p in self._grad_tensors()])
            return ops.realnvp_sample_tape(theta, eps.contiguous().float(), packed, self._param_list(), D, K, W)
        return self.native_sample(eps)

    def sample(self, shape: Tuple) -> torch.Tensor:
        return self.sample_and_log_prob(shape)[0]

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        _ops.require_device(x, "x")
        if torch.is_grad_enabled() and (x.requires_grad or self._params_need_grad()):
            # the differentiable custom op: forward = HIP density + tape, backward (registered with
            # torch.library.register_autograd in _ops.py) = fabhip::realnvp_param_grad.  `theta` is the autograd handle
            # of the parameters: FlatAdam's single leaf, else the concatenation of the nn.Parameters.
            ops = _ops.load()
            packed, D, K, W = self.native(need_inverse=False)
            theta = self._flat_leaf if self._flat_leaf is not None else \
                torch.cat([p.reshape(-1) for p in self._grad_tensors()])
            xd = x.contiguous().float()
            return ops.realnvp_logprob_tape(theta, xd, packed, self._param_list(), D, K, W, bool(x.requires_grad))[0]
        return self.native_log_prob(x)[0]

    # ---- training path: HIP forward with a tape + parameter-gradient GEMMs (csrc/train_kernels.hip) -----
    def _grad_tensors(self):
        """Parameters in the order of the flat gradient image (fabhip_flow_grad_layout)."""
        out = []
        for l1, l2, l3, aff in self._layers():
            out += [l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, aff.L, aff.U, aff.log_S]
        q0 = self._nf_model.q0
        out += [q0.loc, q0.log_scale]
        for an in self._act_norms():
            out += [an.s, an.t]
        return out

    def _grad_views(self, flat: torch.Tensor):
        """Views of the flat gradient image, one per tensor of `_grad_tensors()`."""
        if self._grad_layout is None:
            self._grad_layout = [int(v) for v in _ops.load().flow_grad_layout(self.dim, self.n_layers, self.width)]
        stride, w1, b1, w2, b2, w3, b3, oL, oU, oS, loc, lsc, _, an_base, _ = self._grad_layout
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
        if self.act_norm:
            for k in range(self.n_layers):
                o = an_base + 2 * D * k
                views += [flat[o: o + D].view(1, D), flat[o + D: o + 2 * D].view(1, D)]
        return views

    def grad_floats(self) -> int:
        """Length of the flat gradient / parameter image (fabhip_flow_grad_layout)."""
        if self._grad_layout is None:
            self._grad_layout = [int(v) for v in _ops.load().flow_grad_layout(self.dim, self.n_layers, self.width)]
        return self._grad_layout[14] if self.act_norm else self._grad_layout[12]

    # ---- native (HIP) entry points: torch.ops.fabhip.* --------------------------------------------------------------
    def _params_need_grad(self):
        return any(p.requires_grad for p in self.parameters())

    def _layers(self):
        fl, st = self._nf_model.flows, (3 if self.act_norm else 2)
        for i in range(self.n_layers):
            net = fl[st * i].flows[1].param_map.net
            yield net[0], net[2], net[4], fl[st * i + 1]

    def _act_norms(self):
        return [self._nf_model.flows[3 * i + 2] for i in range(self.n_layers)] if self.act_norm else []

    def _ensure_act_norm(self):
        if self.act_norm and not self._act_norm_ready:
            if any(float(an.data_dep_init_done) <= 0 for an in self._act_norms()):
                self.init_act_norm()
            self._act_norm_ready = True                  # (checked once: the flags only ever go 0 -> 1)

    @torch.no_grad()
    def init_act_norm(self, n_samples: int = 500, eps: torch.Tensor = None):
        """Data-dependent initialisation of the ActNorm layers, what the reference's builder triggers with
        `wrapped_dist.sample((500,))` (make_normflow_model.py:94-95 -> ActNorm.forward on its first batch): layer by
        layer, s = -log(std(z) + 1e-6), t = -mean(z) exp(s) of the batch that reaches the layer (unbiased std, like
        torch.std).  Every partial flow is sampled by the HIP kernel (layers above the one being initialised are
        skipped by sampling a flow truncated to the first k + 1 layers)."""
        if not self.act_norm:
            return
        dev = self._nf_model.q0.loc.device
        if eps is None:
            eps = torch.randn(n_samples, self.dim, device=dev)
        ops = _ops.load()
        for k, an in enumerate(self._act_norms()):
            if float(an.data_dep_init_done) > 0:
                continue
            an.s.zero_(); an.t.zero_()
            K = k + 1
            tensors = self._param_list(n_layers=K)
            packed = torch.empty(ops.flow_packed_floats(self.dim, K, self.width), dtype=torch.float32, device=dev)
            ops.realnvp_pack([t.detach().contiguous().float() for t in tensors], self.dim, K, self.width, True, packed)
            z, _ = ops.realnvp_sample(packed, self.dim, K, self.width, eps.contiguous().float())
            s = -torch.log(z.std(dim=0, keepdim=True) + 1e-6)
            an.s.copy_(s)
            an.t.copy_(-z.mean(dim=0, keepdim=True) * torch.exp(s))
            an.data_dep_init_done.fill_(1.0)

    def _param_list(self, n_layers: int = None):
        """`Tensor[] params` of the ops: per layer {w1, b1, w2, b2, w3, b3, L, U, log_S, sign_S, P}, then loc, log_scale,
        then (act_norm) one {s, t} pair per layer.  n_layers: only the first layers (init_act_norm)."""
        K = self.n_layers if n_layers is None else n_layers
        tensors = []
        for l1, l2, l3, aff in list(self._layers())[:K]:
            tensors += [l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, aff.L, aff.U, aff.log_S,
                        aff.sign_S, aff.P]
        q0 = self._nf_model.q0
        tensors += [q0.loc, q0.log_scale]
        for an in self._act_norms()[:K]:
            tensors += [an.s.reshape(-1), an.t.reshape(-1)]
        return tensors

    _LEAF_ATTRS = (("weight", "bias"), ("weight", "bias"), ("weight", "bias"), ("L", "U", "log_S", "sign_S", "P"))

    def _param_list_fast(self, for_key: bool = False):
        """`_param_list()` for the per-call currency check of `native()`: the leaf MODULES are looked up once (the module tree
        of a flow is fixed after construction; a replaced `_nf_model` rebuilds the cache), their tensors are read from the
        modules' parameter / buffer dicts on every call (a re-assigned Parameter is seen).  ~10x cheaper than walking
        `nn.Module.__getattr__` 200 times per AIS call while the GPU waits for the host."""
        nf = self._nf_model
        c = self.__dict__.get("_leaf_cache")
        if c is None or c[0] is not nf or c[1] != (self.n_layers, self.act_norm):
            c = (nf, (self.n_layers, self.act_norm), list(self._layers()), self._act_norms(), nf.q0)
            self.__dict__["_leaf_cache"] = c
        tensors = []
        for mods in c[2]:
            for m, names in zip(mods, self._LEAF_ATTRS):
                pr, bf = m._parameters, m._buffers
                for n in names:
                    t = pr.get(n)
                    tensors.append(t if t is not None else bf[n])
        q0 = c[4]
        tensors.append(q0._parameters["loc"] if "loc" in q0._parameters else q0.loc)
        tensors.append(q0._parameters["log_scale"] if "log_scale" in q0._parameters else q0.log_scale)
        for an in c[3]:
            tensors += [an.s, an.t] if for_key else [an.s.reshape(-1), an.t.reshape(-1)]
        return tensors

    def invalidate_native(self):
        """Forget the registered parameter set (`_param_key`).  Since round 5 `_param_key` compares every registered tensor object
        with what its module holds, so a re-assigned Parameter / buffer (`layer.weight = nn.Parameter(..)`,
        `load_state_dict(assign=True)`) re-registers by itself; this stays as the explicit form (and for a replaced sub-MODULE)."""
        self.__dict__.pop("_pset", None)                       # (the slot of the op layer, `_pset_handle`, is kept and re-used)
        self.__dict__.pop("_leaf_cache", None)
        self._packed_key = None

    def _own_handle(self):
        h = self.__dict__.get("_pset_handle")                  # (owner token, slot): a deep copy / un-pickled flow carries its
        return h[1] if h is not None and h[0] == _ops.owner_token(self) else -1    # source's entry and must not use it

    def __getstate__(self):
        """Pickling / copy.deepcopy: the handles of the op layer's parameter-set registry (and the caches built on them) belong to
        THIS object in THIS process and do not travel - a copy or an un-pickled flow registers its own tensors (the
        owner token alone, (process nonce, id), can be met again by an object created after its source was collected)."""
        st = dict(self.__dict__)
        for k in ('_pset', '_pset_handle', '_leaf_cache'):
            st.pop(k, None)
        return st

    def __del__(self):
        try:
            h = self._own_handle()
            if h >= 0:
                _ops.load().tensors_key_release(h)             # the op layer drops its references to this flow's tensors
        except Exception:                                      # noqa: BLE001 (interpreter shutdown)
            pass

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self.invalidate_native()                                # (conversions may replace parameter objects under a torch.__future__ flag)
        return out

    def _param_key(self, ops):
        """Identity of the current parameter values: the tensors are registered with the op layer ONCE (112 dict look-ups and a
        112-tensor list through the dispatcher cost 35 us per AIS call while the GPU has nothing to do); per call, every registered
        object is compared by identity with what its module holds now (~4 us) and one integer goes through the dispatcher."""
        nf = self._nf_model
        c = self.__dict__.get("_pset")
        if c is not None and c[0] is nf and c[1] == (self.n_layers, self.act_norm) and c[2] == self._own_handle():
            ok = True
            for d, n, obj in c[3]:
                if d.get(n) is not obj:
                    ok = False
                    break
            if ok:
                return tuple(ops.tensors_key_of(c[2]))
        # (the Parameter / buffer OBJECTS themselves, not detached aliases: `param.data = ...` acts on the registered TensorImpl)
        self.__dict__.pop("_leaf_cache", None)                 # (a copy's cache names the source's modules)
        handle = ops.tensors_key_register(self._param_list_fast(for_key=True), self._own_handle())
        self.__dict__["_pset_handle"] = (_ops.owner_token(self), handle)
        # identity probes for EVERY registered tensor: (owning dict, name, object).  ADVICE r4: three probes did not see a
        # Parameter / buffer object re-assigned in another layer (`layer.weight = nn.Parameter(..)`, `load_state_dict(assign=True)`),
        # and the stale image was then sampled silently; 112 dict look-ups cost ~4 us per call
        lc = self.__dict__["_leaf_cache"]
        probes = []
        for mods in lc[2]:
            for m, names in zip(mods, self._LEAF_ATTRS):
                for n in names:
                    d = m._parameters if m._parameters.get(n) is not None else m._buffers
                    probes.append((d, n, d[n]))
        q0 = lc[4]
        for n in ("loc", "log_scale"):
            if q0._parameters.get(n) is not None:
                probes.append((q0._parameters, n, q0._parameters[n]))
        for an in lc[3]:
            for n in ("s", "t"):
                d = an._parameters if an._parameters.get(n) is not None else an._buffers
                if d.get(n) is not None:
                    probes.append((d, n, d[n]))
        self.__dict__["_pset"] = (nf, (self.n_layers, self.act_norm), handle, probes)
        return tuple(ops.tensors_key_of(handle))

    def native(self, need_inverse: bool = True):
        """(packed image, dim, n_layers, width) - the flow arguments of the ops; the image is re-tiled by the pack
        kernels whenever a parameter changed.  need_inverse=False (density evaluations only, e.g. the minibatch loop of
        the trainer) skips the W^-1 matrices; the next caller that samples gets a full re-pack."""
        ops = _ops.load()
        q0 = self._nf_model.q0
        _ops.require_device(q0.loc, "RealNVP parameters")
        self._ensure_act_norm()
        key = self._param_key(ops)                             # (storage address, version) of every tensor, mixed in C++
        if key != self._packed_key or (need_inverse and not self._packed_has_inverse):
            tensors = self._param_list_fast()
            n = ops.flow_packed_floats(self.dim, self.n_layers, self.width)
            if n < 0:
                raise _ops.FabhipError(f"flow shape not supported by the kernels: dim={self.dim} width={self.width}")
            if self._packed is None or self._packed.numel() != n or self._packed.device != q0.loc.device:
                self._packed = torch.empty(n, dtype=torch.float32, device=q0.loc.device)
            with torch.no_grad():
                ops.realnvp_pack([t.detach() for t in tensors], self.dim, self.n_layers, self.width, bool(need_inverse),
                                 self._packed)
            self._packed_has_inverse = need_inverse
            self._packed_key = key
            # (how often the image was rebuilt: FlatAdam moves the parameters without touching autograd's version counters and
            #  asks for a re-pack by clearing `_packed_key` - the key alone then repeats; ais.py keys its prefetch on this count)
            self.__dict__["_pack_count"] = self.__dict__.get("_pack_count", 0) + 1
        return self._packed, self.dim, self.n_layers, self.width

    def native_sample(self, eps: torch.Tensor):
        _ops.require_device(eps, "eps")
        fargs = self.native()
        x, log_q = _ops.load().realnvp_sample(*fargs, eps.detach().contiguous().float())
        return x, log_q

    def native_log_prob(self, x: torch.Tensor, with_grad: bool = False):
        _ops.require_device(x, "x")
        fargs = self.native(need_inverse=False)
        log_q, grad = _ops.load().realnvp_logprob_grad(*fargs, x.detach().contiguous().float(), bool(with_grad),
                                                       _ops.precision_of(self))
        return log_q, (grad if with_grad else None)

    # ---- autograd-free training entry points (the same two ops the autograd path runs) -------------------------------
    def log_prob_with_tape(self, x: torch.Tensor, want_grad_x: bool = False):
        """(log q(x), tape handle[, d log q / dx]) through fabhip::realnvp_logprob_tape, no autograd graph."""
        _ops.require_device(x, "x")
        packed, D, K, W = self.native(need_inverse=False)
        xd = x.detach().contiguous().float()
        with torch.no_grad():
            log_q, grad_x, tape = _ops.load().realnvp_logprob_tape(packed, xd, packed, [], D, K, W, bool(want_grad_x))
        handle = (tape, xd.shape[0], self._packed_key)
        return (log_q, handle, grad_x) if want_grad_x else (log_q, handle)

    def param_grad_flat(self, tape_handle, coef: torch.Tensor) -> torch.Tensor:
        """sum_b coef[b] * d log q(x_b) / d theta as one flat gradient image (layout: `_grad_views`)."""
        tape, B, key = tape_handle
        if self._packed_key != key:
            raise _ops.FabhipError("flow parameters were modified between log_prob_with_tape(x) and param_grad_flat()")
        packed, D, K, W = self.native(need_inverse=False)
        c = coef.detach().contiguous().float()
        assert c.shape[0] == B
        with torch.no_grad():
            return _ops.load().realnvp_param_grad([t.detach() for t in self._param_list()], packed, D, K, W, tape, c)

    def log_prob_and_grad(self, x: torch.Tensor):
        """(log q(x), d log q / dx) - what `grad_and_value(x, flow.log_prob)` computes (base.py:50-56)."""
        return self.native_log_prob(x, with_grad=True)


def make_wrapped_normflow_realnvp(dim: int, n_flow_layers: int = 5, layer_nodes_per_dim: int = 10,
                                  act_norm: bool = True) -> RealNVP:
    """Same name/arguments as experiments/make_flow/make_normflow_model.py:82-96.  With act_norm the reference draws 500
    samples at construction to initialise the ActNorm layers (:94-95); here that happens the first time the flow's
    parameters are on the GPU (`RealNVP.init_act_norm`, called lazily by `native()`), the kernels having no CPU path."""
    return RealNVP(dim, n_flow_layers, layer_nodes_per_dim, act_norm)
