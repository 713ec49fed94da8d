"""FlatAdam — the trainer's `clip_grad_norm_` + `torch.optim.Adam.step()` pair
(fab/train_with_prioritised_buffer.py:174-179) as two HIP launches on a flat parameter image.

The flow's parameters are re-pointed (same `nn.Parameter` objects, same state-dict keys) into ONE contiguous
float32 buffer laid out like the flat gradient image of `fabhip_flow_param_grad`; when the gradients of the step
are that image (the usual case: one `flow.log_prob(x)` backward) the step reads it directly, otherwise the
`.grad` tensors are concatenated first.  No host synchronisation: the gradient norm stays on the device and a
non-finite norm skips the update inside the kernel."""
from typing import Optional

import torch

from . import _ops
from .flow import RealNVP


class FlatAdam(torch.optim.Optimizer):
    """`flow`: a fab_torch_amd RealNVP (flat image = the layout of its parameter-gradient kernels, one autograd leaf), or
    any other nn.Module (e.g. the spline flow: parameters in `.parameters()` order, gradients concatenated per step)."""

    def __init__(self, flow: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flow = flow
        self.native = isinstance(flow, RealNVP)
        self._params = flow._grad_tensors() if self.native else [p for p in flow.parameters()]
        for p in self._params:
            _ops.require_device(p, "flow parameters (move the flow to the GPU before building FlatAdam)")
        super().__init__(self._params, dict(lr=lr, betas=betas, eps=eps))
        self.n = flow.grad_floats() if self.native else sum(p.numel() for p in self._params)
        dev = self._params[0].device
        self.theta = torch.empty(self.n, dtype=torch.float32, device=dev)
        if self.native:                              # (+ the ActNorm pairs for an act_norm flow)
            views = flow._grad_views(self.theta)
        else:
            views, o = [], 0
            for p in self._params:
                views.append(self.theta[o:o + p.numel()].view(p.shape))
                o += p.numel()
        with torch.no_grad():
            for p, v in zip(self._params, views):
                v.copy_(p.detach())
                p.data = v                                   # the Parameter now lives inside theta
        self._offsets = [v.data_ptr() - self.theta.data_ptr() for v in views]
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.steps = torch.zeros(1, dtype=torch.int32, device=dev)     # applied steps (device: skips happen there)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        flow._packed_key = None
        if self.native:
            # one autograd leaf for the whole flow: flow.log_prob(x).backward() then delivers ONE flat gradient
            # (theta.grad) instead of 112 per-parameter views (the per-parameter .grad stay None in this mode)
            self.theta.requires_grad_(True)
            flow._flat_leaf = self.theta

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=True)
        self.theta.grad = None

    def _check_alias(self):
        base = self.theta.data_ptr()
        for p, off in ((self._params[0], self._offsets[0]), (self._params[-1], self._offsets[-1])):
            if p.data_ptr() != base + off:
                raise _ops.FabhipError("the flow's parameters were re-allocated after FlatAdam was built "
                                       "(e.g. by .to()/.cuda()); build the optimiser after moving the flow")

    def _flat_grad(self) -> torch.Tensor:
        """theta.grad (flat mode), plus whatever reached the individual parameters through other autograd paths."""
        extra = [p for p in self._params if p.grad is not None]
        if self.theta.grad is not None:
            g = self.theta.grad
            if extra:
                g = g + self._cat_grads()
            return g.contiguous()
        if not extra:
            raise _ops.FabhipError("FlatAdam.step(): parameters have no gradient")
        g0 = self._params[0].grad
        if g0 is None:
            return self._cat_grads()
        base = g0.data_ptr() - self._offsets[0]
        flat = getattr(self.flow, "_last_flat_grad", None)
        if flat is not None and flat.data_ptr() == base and all(
                p.grad is not None and p.grad.data_ptr() == base + off and p.grad.is_contiguous()
                for p, off in zip(self._params, self._offsets)):
            return flat
        return self._cat_grads()

    def _cat_grads(self) -> torch.Tensor:
        zero = None
        parts = []
        for p in self._params:
            if p.grad is None:
                zero = torch.zeros(1, device=self.theta.device) if zero is None else zero
                parts.append(zero.expand(p.numel()))
            else:
                parts.append(p.grad.reshape(-1))
        return torch.cat(parts).float().contiguous()

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None,
             flat_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Clip to `max_grad_norm` (None / inf: no clipping), then Adam.  Returns the gradient norm as a device
        tensor (what clip_grad_norm_ returns); a non-finite norm leaves the parameters untouched.
        `flat_grad`: a gradient image from `RealNVP.param_grad_flat` (no autograd involved), else the .grad fields."""
        assert closure is None
        self._check_alias()
        g = flat_grad if flat_grad is not None else self._flat_grad()
        if g.numel() != self.n or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous() \
                or g.device != self.theta.device:
            raise _ops.FabhipError(f"FlatAdam.step(): gradient image must be a contiguous float32 tensor of "
                                   f"{self.n} elements on {self.theta.device} (got {tuple(g.shape)} {g.dtype} {g.device})")
        grp = self.param_groups[0]
        mx = 0.0 if (max_grad_norm is None or max_grad_norm == float("inf")) else float(max_grad_norm)
        _ops.load().adam_clip_step(self.theta.detach(), g.detach(), self.m, self.v, float(grp["lr"]),
                                   float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), self.steps, mx,
                                   self.grad_norm)
        self.flow._packed_key = None                          # parameters changed behind autograd's version counters
        return self.grad_norm[0]

    def state_dict(self):
        return {"t": int(self.steps.item()), "m": self.m, "v": self.v, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                        for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps.fill_(int(sd["t"]))
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
