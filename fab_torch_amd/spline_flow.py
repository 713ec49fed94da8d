"""Circular / linear-tail rational-quadratic spline coupling flow backed by the HIP kernels of csrc/spline_kernels.hip.

Host-side mirror of what the reference builds with normflows for alanine dipeptide
(experiments/make_flow/make_aldp_model.py:57-71,121-134,146-167, `flow.type = circular-coup-nsf`,
experiments/aldp/config/fab_buff.yaml:20-36) and wraps in fab/wrappers/normflows.py:8-31: n_layers x
`CircularCoupledRationalQuadraticSpline(dim, blocks_per_layer=1, hidden_units, ind_circ, tail_bound, num_bins=8,
init_identity, mask)` with alternating random binary masks, a `PeriodicShift` after every second layer, a final
`PeriodicWrap`, base `UniformGaussian`.  Same `Distribution` methods (fab/types_.py:8-27) and the same state-dict key
names as the normflows modules (`_nf_model.flows.{i}.prqct.transform_net.{initial_layer, blocks.0.linear_layers.{0,1},
final_layer, preprocessing.weights}`, `...prqct.unconditional_transform.unnormalized_{widths,heights,derivatives}`) so
that a normflows checkpoint of this architecture loads by key (fab/core.py:237-240) - the "checkpoint importer" of
SURVEY.md section 8f-4.  normflows is absent from the reference tree: the arithmetic is specified by oracle/spline.py
(parity unpinned by the reference).

`log_prob`, its gradient w.r.t. x and `sample_and_log_prob` run on the GPU through torch.ops.fabhip.spline_*; the
sampler drives this flow through the generic plug-in path of the transition operators (`log_prob` is differentiable
w.r.t. x through a custom autograd Function whose backward is the kernels' own reverse sweep).

Training (`loss.backward()` through `log_prob`, the forward-KL term of the FAB loss, fab/core.py:114-127 /
fab/train_with_prioritised_buffer.py:160-171): when a parameter requires grad, `log_prob` runs
`fabhip::spline_logprob_tape`, which also writes the conditioner activations and the reverse sweep's cotangents of every
layer to a tape; the backward turns them into the parameter gradients with one rocBLAS GEMM per Linear (torch.mm over
tape slices, the upstream coefficient folded into the cotangent rows) - no ATen re-implementation of the flow exists.
The sampling direction is not differentiable w.r.t. the parameters (the reverse-KL style baseline losses are built for
the RealNVP family only)."""
import math
from typing import Sequence, Tuple

import torch
import torch.nn as nn

from . import _ops

NUM_BINS = 8
MIN_DERIVATIVE = 1e-3


class _PeriodicFeatures(nn.Module):
    def __init__(self, ind, scale):
        super().__init__()
        self.register_buffer("ind", torch.as_tensor(list(ind), dtype=torch.long))
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.float32).reshape(-1))
        self.weights = nn.Parameter(torch.ones(len(ind), 2))


class _ResidualBlock(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.linear_layers = nn.ModuleList([nn.Linear(features, features), nn.Linear(features, features)])
        nn.init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
        nn.init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)


class _ResidualNet(nn.Module):
    def __init__(self, in_features, out_features, hidden, preprocessing):
        super().__init__()
        self.hidden_features = hidden
        self.preprocessing = preprocessing
        self.initial_layer = nn.Linear(in_features, hidden)
        self.blocks = nn.ModuleList([_ResidualBlock(hidden)])
        self.final_layer = nn.Linear(hidden, out_features)


class _UnconditionalRQS(nn.Module):
    def __init__(self, n, circ, tail_bound):
        super().__init__()
        self.register_buffer("circ", circ)
        self.register_buffer("tail_bound", tail_bound)
        const = math.log(math.exp(1 - MIN_DERIVATIVE) - 1)
        self.unnormalized_widths = nn.Parameter(torch.zeros(n, NUM_BINS))
        self.unnormalized_heights = nn.Parameter(torch.zeros(n, NUM_BINS))
        self.unnormalized_derivatives = nn.Parameter(const * torch.ones(n, NUM_BINS + 1))


class _PRQCT(nn.Module):
    def __init__(self, mask, hidden, circ_all, tail_bound_all, init_identity):
        super().__init__()
        feats = torch.arange(mask.shape[0])
        self.register_buffer("identity_features", feats[mask <= 0])
        self.register_buffer("transform_features", feats[mask > 0])
        idf, trf = self.identity_features, self.transform_features
        circ_id = [i for i, f in enumerate(idf.tolist()) if bool(circ_all[f])]
        pf = _PeriodicFeatures(circ_id, math.pi / tail_bound_all[idf][circ_id]) if circ_id else None
        self.transform_net = _ResidualNet(len(idf), len(trf) * (3 * NUM_BINS + 1), hidden, pf)
        if init_identity:
            nn.init.constant_(self.transform_net.final_layer.weight, 0.0)
            nn.init.constant_(self.transform_net.final_layer.bias, math.log(math.exp(1 - MIN_DERIVATIVE) - 1))
        self.register_buffer("circ_t", circ_all[trf].clone())
        self.register_buffer("tb_t", tail_bound_all[trf].clone())
        self.unconditional_transform = _UnconditionalRQS(len(idf), circ_all[idf].clone(), tail_bound_all[idf].clone())


class _Coupling(nn.Module):
    def __init__(self, mask, hidden, circ_all, tail_bound_all, init_identity):
        super().__init__()
        self.prqct = _PRQCT(mask, hidden, circ_all, tail_bound_all, init_identity)


class _PeriodicShift(nn.Module):
    def __init__(self, ind, bound, shift):
        super().__init__()
        self.register_buffer("ind", torch.as_tensor(list(ind), dtype=torch.long))
        self.register_buffer("bound", torch.as_tensor(bound, dtype=torch.float32).reshape(-1))
        self.register_buffer("shift", torch.as_tensor(shift, dtype=torch.float32).reshape(-1))


class _PeriodicWrap(nn.Module):
    def __init__(self, ind, bound):
        super().__init__()
        self.register_buffer("ind", torch.as_tensor(list(ind), dtype=torch.long))
        self.register_buffer("bound", torch.as_tensor(bound, dtype=torch.float32).reshape(-1))


class _UniformGaussian(nn.Module):
    def __init__(self, ndim, ind_circ, scale):
        super().__init__()
        self.shape = (ndim,)
        circ = torch.zeros(ndim, dtype=torch.bool)
        circ[list(ind_circ)] = True
        self.register_buffer("circ", circ)
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.float32).reshape(-1))


class _NormalizingFlow(nn.Module):
    def __init__(self, q0, flows):
        super().__init__()
        self.q0 = q0
        self.flows = nn.ModuleList(flows)


class _SplineLogProb(torch.autograd.Function):
    """log q(x) with d log q / dx from the kernels' reverse sweep: what `grad_and_value(x, flow.log_prob)`
    (fab/sampling_methods/base.py:50-56) differentiates through on the generic transition path."""

    @staticmethod
    def forward(ctx, flow, x):
        lq, g = flow.native_log_prob(x, with_grad=True)
        ctx.save_for_backward(g)
        return lq

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return None, grad_out[:, None] * g


class _SplineLogProbTape(torch.autograd.Function):
    """log q(x) with gradients w.r.t. x AND the flow parameters (`params` = flow._train_params(), passed only so that
    autograd routes the gradients to them)."""

    @staticmethod
    def forward(ctx, flow, x, *params):
        xd = x.detach().contiguous().float()
        lq, gx, tape = _ops.load().spline_logprob_tape(*flow.native(), xd)
        ctx.flow, ctx.B = flow, xd.shape[0]
        ctx.save_for_backward(gx, tape)
        return lq

    @staticmethod
    def backward(ctx, grad_out):
        gx, tape = ctx.saved_tensors
        c = grad_out.detach().float().contiguous()
        grads = ctx.flow._param_grads_from_tape(tape, c, ctx.B)
        return (None, c[:, None] * gx if ctx.needs_input_grad[1] else None, *grads)


class _SplineSampleTape(torch.autograd.Function):
    """x, log_q = flow.sample_and_log_prob() with gradients w.r.t. the flow parameters and the base noise (the reparameterised
    baseline losses, fab/core.py:130-152).  Forward = the HIP sampler.  Backward = `fabhip::spline_sample_vjp_tape` (include/fabhip.h,
    fabhip_spline_sample_vjp_tape): with S the log_prob direction (x = S^-1(z0; theta)), d/d theta = g_lq d log q(x)/d theta |_x -
    v^T dS/d theta, v = (dS/dx)^-T (g_x + g_lq d log q/dx) - two tapes in the density tape's layout, turned into parameter gradients
    by the same tape GEMMs (coefficients g_lq and 1)."""

    @staticmethod
    def forward(ctx, flow, u, eps, *params):
        ud, ed = u.detach().contiguous().float(), eps.detach().contiguous().float()
        x, lq = _ops.load().spline_sample(*flow.native(), ud, ed)
        ctx.flow, ctx.B = flow, x.shape[0]
        ctx.save_for_backward(ud, ed)
        return x, lq

    @staticmethod
    def backward(ctx, gx, glq):
        ud, ed = ctx.saved_tensors
        flow, B = ctx.flow, ctx.B
        gx = None if gx is None else gx.detach().float().contiguous()
        c = None if glq is None else glq.detach().float().contiguous()
        if gx is None and c is None:
            return (None,) * (3 + len(ctx.needs_input_grad[3:]))
        tape1, tape2, v_base = _ops.load().spline_sample_vjp_tape(*flow.native(), ud, ed, gx, c)
        grads = [None] * len(ctx.needs_input_grad[3:])
        if any(ctx.needs_input_grad[3:]):
            grads = flow._param_grads_from_tape(tape2, torch.ones(B, device=ud.device), B)
            if c is not None:
                grads = [a + b for a, b in zip(grads, flow._param_grads_from_tape(tape1, c, B))]
        g_noise = v_base * flow._nf_model.q0.scale if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        circ = flow._circ
        g_u = g_noise * circ if ctx.needs_input_grad[1] else None
        g_eps = g_noise * (~circ) if ctx.needs_input_grad[2] else None
        return (None, g_u, g_eps, *grads)


class CircularCoupledRQSFlow(nn.Module):
    def __init__(self, dim: int, n_layers: int, hidden_units: int, ind_circ: Sequence[int], tail_bound,
                 num_bins: int = 8, blocks_per_layer: int = 1, seed: int = 0, circ_shift: str = "random",
                 init_identity: bool = True):
        super().__init__()
        if num_bins != NUM_BINS or blocks_per_layer != 1:
            raise NotImplementedError("the HIP spline kernels are built for 8 bins and one residual block per conditioner "
                                      "(experiments/aldp/config/fab_buff.yaml:28-32)")
        self.dim, self.n_layers, self.hidden = dim, n_layers, hidden_units
        tail_bound = torch.as_tensor(tail_bound, dtype=torch.float32).reshape(-1)
        if tail_bound.numel() == 1:
            tail_bound = tail_bound.expand(dim).clone()
        ind_circ = list(ind_circ)
        circ = torch.zeros(dim, dtype=torch.bool)
        circ[ind_circ] = True
        bound_circ = tail_bound[ind_circ]
        scale = torch.ones(dim)
        scale[ind_circ] = 2 * bound_circ
        layers, mask = [], None
        for i in range(n_layers):
            if i % 2 == 0:                                 # nf.utils.masks.create_random_binary_mask(ndim, seed=seed + i)
                g = torch.Generator().manual_seed(seed + i)
                mask = torch.zeros(dim)
                mask[torch.multinomial(torch.ones(dim), dim // 2 + dim % 2, replacement=False, generator=g)] += 1
            else:
                mask = 1 - mask
            layers.append(_Coupling(mask.clone(), hidden_units, circ, tail_bound, init_identity))
            if i % 2 == 1 and i != n_layers - 1 and circ_shift is not None and ind_circ:
                if circ_shift == "constant":
                    layers.append(_PeriodicShift(ind_circ, bound_circ, bound_circ))
                else:
                    g = torch.Generator().manual_seed(seed + i)
                    layers.append(_PeriodicShift(ind_circ, bound_circ, (torch.rand([], generator=g) + 0.5) * bound_circ))
        if ind_circ:
            layers.append(_PeriodicWrap(ind_circ, bound_circ))
        self._nf_model = _NormalizingFlow(_UniformGaussian(dim, ind_circ, scale), layers)
        self.precision = None             # None / "fp32" / "fast": see RealNVP.precision (the density + gradient kernel)
        self.register_buffer("_circ", circ)
        self.register_buffer("_tail_bound", tail_bound.clone())
        self._packed = None
        self._packed_key = None

    # ---- Distribution interface (fab/types_.py:8-27) ---------------------------------------------------------------
    @property
    def event_shape(self) -> Tuple[int, ...]:
        return self._nf_model.q0.shape

    def sample_and_log_prob(self, shape: Tuple[int, ...], u: torch.Tensor = None, eps: torch.Tensor = None):
        assert len(shape) == 1
        dev = self._tail_bound.device
        if u is None:
            u = torch.rand((shape[0], self.dim), dtype=torch.float32, device=dev)
        if eps is None:
            eps = torch.randn((shape[0], self.dim), dtype=torch.float32, device=dev)
        _ops.require_device(u, "u")
        if torch.is_grad_enabled():
            params = self._train_params()
            if u.requires_grad or eps.requires_grad or any(p.requires_grad for p in params):
                return _SplineSampleTape.apply(self, u, eps, *params)
        x, log_q = _ops.load().spline_sample(*self.native(), u.contiguous().float(), eps.contiguous().float())
        return x, log_q

    def sample(self, shape: Tuple) -> torch.Tensor:
        return self.sample_and_log_prob(shape)[0]

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        _ops.require_device(x, "x")
        if torch.is_grad_enabled():
            params = self._train_params()
            if any(p.requires_grad for p in params):
                return _SplineLogProbTape.apply(self, x, *params)
            if x.requires_grad:
                return _SplineLogProb.apply(self, x)
        return self.native_log_prob(x)[0]

    # ---- training: parameter gradients from the kernels' tape ---------------------------------------------------------
    def _train_params(self):
        """The trainable tensors in the order `_param_grads_from_tape` returns their gradients."""
        out = []
        for f, _, _ in self._structure():
            net, u = f.prqct.transform_net, f.prqct.unconditional_transform
            out += [net.initial_layer.weight, net.initial_layer.bias,
                    net.blocks[0].linear_layers[0].weight, net.blocks[0].linear_layers[0].bias,
                    net.blocks[0].linear_layers[1].weight, net.blocks[0].linear_layers[1].bias,
                    net.final_layer.weight, net.final_layer.bias]
            if net.preprocessing is not None:
                out.append(net.preprocessing.weights)
            out += [u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives]
        return out

    def _param_grads_from_tape(self, tape: torch.Tensor, c: torch.Tensor, B: int):
        """d (sum_b c_b log q(x_b)) / d parameter for every tensor of `_train_params()` (tape layout: include/fabhip.h,
        fabhip_spline_tape_layout)."""
        lay = _ops.load().spline_tape_layout(self.dim, self.n_layers, self.hidden, B)
        stride, (o_xi, o_a0, o_da0, o_r0, o_r1, o_h1, o_dh1, o_dt, o_dh0, o_dp, o_du), Wp, NFP, UW = \
            lay[1], lay[2:13], lay[13], lay[14], lay[15]
        W, NP, L = self.hidden, 3 * NUM_BINS + 1, self.n_layers
        # all layers at once: the tape holds the same matrices at the same offsets for every layer (padding columns of
        # the activations / cotangents are zero), so each Linear's weight gradient is ONE batched GEMM over the layers
        T = tape[:L * stride].view(L, stride)

        def m3(off, width):
            return T[:, off: off + B * width].view(L, B, width)
        cc = c.view(1, B, 1)
        # weight AND bias gradient of each Linear: ONE launch of the batch-reduction GEMM kernel over all layers
        # (fabhip_tape_gemm: coefficients folded into the cotangent operand, bias = its column sums)
        tg = _ops.load().tape_gemm
        gw0, gb0 = tg(tape, stride, L, o_dh0, Wp, W, o_a0, 64, 64, c, True)       # [L, W, 64], [L, W]
        gwa, gba = tg(tape, stride, L, o_dt, Wp, W, o_r0, Wp, W, c, True)
        gwb, gbb = tg(tape, stride, L, o_dh1, Wp, W, o_r1, Wp, W, c, True)
        gwf, gbf = tg(tape, stride, L, o_dp, NFP, NFP, o_h1, Wp, W, c, True)      # [L, NFP, W], [L, NFP]
        du = (cc * m3(o_du, UW)).sum(1)                                            # [L, 64 * 25] (rows >= n_id: unwritten)
        grads = []
        for l, (f, _, _) in enumerate(self._structure()):
            pr, net = f.prqct, f.prqct.transform_net
            n_id, n_tr = len(pr.identity_features), len(pr.transform_features)
            nout = n_tr * NP
            grads += [gw0[l, :, :n_id], gb0[l], gwa[l], gba[l], gwb[l], gbb[l], gwf[l, :nout], gbf[l, :nout]]
            if net.preprocessing is not None:
                pf = net.preprocessing
                xs = m3(o_xi, 64)[l][:, pf.ind] * pf.scale
                da = (c[:, None] * m3(o_da0, 64)[l])[:, pf.ind]
                grads.append(torch.stack([(da * torch.sin(xs)).sum(0), (da * torch.cos(xs)).sum(0)], dim=1))
            dul = du[l, :n_id * NP].view(n_id, NP)
            grads += [dul[:, :NUM_BINS], dul[:, NUM_BINS:2 * NUM_BINS], dul[:, 2 * NUM_BINS:]]
        return grads

    def log_prob_and_grad(self, x: torch.Tensor):
        return self.native_log_prob(x, with_grad=True)

    def native_log_prob(self, x: torch.Tensor, with_grad: bool = False):
        _ops.require_device(x, "x")
        lq, g = _ops.load().spline_logprob_grad(*self.native(), x.detach().contiguous().float(), bool(with_grad),
                                                _ops.precision_of(self))
        return lq, (g if with_grad else None)

    # ---- packing -----------------------------------------------------------------------------------------------------
    def _structure(self):
        """[(coupling module, pre_shift or None, pre_wrap, post_shift or None)] in layer order."""
        out = []
        flows = list(self._nf_model.flows)
        couplings = [(i, f) for i, f in enumerate(flows) if isinstance(f, _Coupling)]
        for n, (i, f) in enumerate(couplings):
            nxt = flows[i + 1] if i + 1 < len(flows) else None
            post = nxt if isinstance(nxt, _PeriodicShift) else None       # PeriodicShift right after this coupling
            is_last = n == len(couplings) - 1
            wrap = flows[-1] if (is_last and isinstance(flows[-1], _PeriodicWrap)) else None
            out.append((f, post, wrap))
        return out

    def _param_list(self):
        dev = self._tail_bound.device
        D = self.dim
        tensors = []
        cache = self.__dict__.setdefault("_meta_cache", {})          # static structure: built once per device
        for li, (f, post, wrap) in enumerate(self._structure()):
            pr = f.prqct
            net = pr.transform_net
            key = (li, str(dev))
            if key in cache:
                pr, net, u = f.prqct, f.prqct.transform_net, f.prqct.unconditional_transform
                tensors += [cache[key], net.initial_layer.weight, net.initial_layer.bias,
                            net.blocks[0].linear_layers[0].weight, net.blocks[0].linear_layers[0].bias,
                            net.blocks[0].linear_layers[1].weight, net.blocks[0].linear_layers[1].bias,
                            net.final_layer.weight, net.final_layer.bias,
                            net.preprocessing.weights if net.preprocessing is not None else cache[("empty", str(dev))],
                            u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives]
                continue
            idf, trf = pr.identity_features.tolist(), pr.transform_features.tolist()
            meta = torch.zeros(12, 64)
            meta[0, :] = -1; meta[1, :] = -1
            meta[0, :len(idf)] = torch.tensor(idf, dtype=torch.float32)
            meta[1, :len(trf)] = torch.tensor(trf, dtype=torch.float32)
            meta[2, :D] = self._circ.float().cpu()
            meta[3, :D] = self._tail_bound.cpu()
            n_pf = 0
            if net.preprocessing is not None:
                pf = net.preprocessing
                for k, i in enumerate(pf.ind.tolist()):
                    meta[4, i] = 1.0; meta[5, i] = float(pf.scale[k]); meta[6, i] = float(k)
                n_pf = len(pf.ind)
            # log_prob direction, BEFORE this layer: the inverse of the PeriodicShift that follows it / the final wrap
            if post is not None:
                meta[7, post.ind.cpu()] = post.shift.cpu().expand(len(post.ind)); meta[8, post.ind.cpu()] = 1.0
                meta[9, post.ind.cpu()] = post.shift.cpu().expand(len(post.ind)); meta[10, post.ind.cpu()] = 1.0
            if wrap is not None:
                meta[8, wrap.ind.cpu()] = 1.0                         # shift 0: PeriodicWrap.inverse
            meta[11, 0], meta[11, 1], meta[11, 2] = len(idf), len(trf), n_pf
            u = pr.unconditional_transform
            cache[key] = meta.to(dev)
            cache.setdefault(("empty", str(dev)), torch.zeros(0, device=dev))
            tensors += [cache[key], net.initial_layer.weight, net.initial_layer.bias,
                        net.blocks[0].linear_layers[0].weight, net.blocks[0].linear_layers[0].bias,
                        net.blocks[0].linear_layers[1].weight, net.blocks[0].linear_layers[1].bias,
                        net.final_layer.weight, net.final_layer.bias,
                        net.preprocessing.weights if net.preprocessing is not None else cache[("empty", str(dev))],
                        u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives]
        q0 = self._nf_model.q0
        return tensors + [q0.scale, q0.circ.float()]

    def invalidate_native(self):
        """Forget the registered parameter / buffer sets (`_param_keys`): needed only after ASSIGNING a new Parameter / buffer
        object to a sub-module (in-place updates, `.data = ...`, `.to()`, `load_state_dict` keep the objects and are seen)."""
        self.__dict__.pop("_pset", None)
        self._packed_key = None

    def _own_handles(self):
        h = self.__dict__.get("_pset_handles")                 # (owner token, (slot, slot)): a deep copy / un-pickled flow carries
        return h[1] if h is not None and h[0] == _ops.owner_token(self) else (-1, -1)    # its source's entry and must not use it

    def __getstate__(self):
        """Pickling / copy.deepcopy: the handles of the op layer's parameter-set registry (and the caches built on them) belong to
        THIS object in THIS process and do not travel - a copy or an un-pickled flow registers its own tensors (the
        owner token alone, (process nonce, id), can be met again by an object created after its source was collected)."""
        st = dict(self.__dict__)
        for k in ('_pset', '_pset_handles'):
            st.pop(k, None)
        return st

    def __del__(self):
        try:
            for h in self._own_handles():
                if h >= 0:
                    _ops.load().tensors_key_release(h)         # the op layer drops its references to this flow's tensors
        except Exception:                                      # noqa: BLE001 (interpreter shutdown)
            pass

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self.invalidate_native()
        return out

    def _param_keys(self, ops):
        """((storage address, version) mix of the parameters, the same of the buffers): the tensor objects are registered with
        the op layer once (fabhip::tensors_key_register), all of them are compared by identity per call, two integers go
        through the dispatcher - instead of walking `parameters()` / `buffers()` (hundreds of attribute reads per AIS call)."""
        c = self.__dict__.get("_pset")
        if c is not None and (c[0], c[1]) != self._own_handles():
            c = None
        if c is not None:
            for d, n, obj in c[2]:
                if d.get(n) is not obj:
                    c = None
                    break
        if c is None:
            plist, blist, seen = [], [], set()
            for m in self.modules():
                for n, t in m._parameters.items():
                    if t is not None and id(t) not in seen:
                        seen.add(id(t)); plist.append((m._parameters, n, t))
                for n, t in m._buffers.items():
                    if t is not None and id(t) not in seen:
                        seen.add(id(t)); blist.append((m._buffers, n, t))
            old = self._own_handles()
            hp = ops.tensors_key_register([e[2] for e in plist], old[0])
            hb = ops.tensors_key_register([e[2] for e in blist], old[1])
            self.__dict__["_pset_handles"] = (_ops.owner_token(self), (hp, hb))
            probes = plist + blist                               # EVERY registered tensor (three probes missed a
                                                                 # Parameter / buffer re-assigned in another layer; ~10 us per call)
            c = (hp, hb, probes)
            self.__dict__["_pset"] = c
        return tuple(ops.tensors_key_of(c[0])), tuple(ops.tensors_key_of(c[1]))

    def native(self):
        """(packed image, dim, n_layers, hidden): the flow arguments of torch.ops.fabhip.spline_*; re-packed whenever a
        parameter changed."""
        ops = _ops.load()
        _ops.require_device(self._tail_bound, "spline flow parameters")
        pkey, bkey = self._param_keys(ops)
        if bkey != self.__dict__.get("_bkey"):                 # masks / shifts / bounds changed (e.g. load_state_dict)
            self.__dict__["_meta_cache"] = {}
            self.__dict__["_bkey"] = bkey
        key = pkey + bkey
        if key != self._packed_key:
            n = ops.spline_packed_floats(self.dim, self.n_layers, self.hidden)
            if n < 0:
                raise _ops.FabhipError(f"spline flow shape not supported: dim={self.dim} hidden={self.hidden}")
            if self._packed is None or self._packed.numel() != n or self._packed.device != self._tail_bound.device:
                self._packed = torch.empty(n, dtype=torch.float32, device=self._tail_bound.device)
            with torch.no_grad():
                ops.spline_pack([t.detach().contiguous().float() for t in self._param_list()], self.dim, self.n_layers,
                                self.hidden, self._packed)
            self._packed_key = key
        return self._packed, self.dim, self.n_layers, self.hidden


def make_wrapped_normflow_spline(dim: int, n_layers: int, hidden_units: int, ind_circ: Sequence[int], tail_bound,
                                 seed: int = 0, circ_shift: str = "random", init_identity: bool = True
                                 ) -> CircularCoupledRQSFlow:
    """The 'circular-coup-nsf' branch of experiments/make_flow/make_aldp_model.py:121-134 behind the `Distribution`
    interface of fab/wrappers/normflows.py."""
    return CircularCoupledRQSFlow(dim, n_layers, hidden_units, ind_circ, tail_bound, seed=seed, circ_shift=circ_shift,
                                  init_identity=init_identity)
