"""`Point` record of the AIS chain — same fields / indexing semantics as
fab/sampling_methods/base.py:7-47 (device tensors)."""
from typing import Optional

import torch


class Point:
    def __init__(self, x: torch.Tensor, log_q: torch.Tensor, log_p: torch.Tensor,
                 grad_log_q: Optional[torch.Tensor] = None, grad_log_p: Optional[torch.Tensor] = None):
        self.x, self.log_q, self.log_p = x, log_q, log_p
        self.grad_log_q, self.grad_log_p = grad_log_q, grad_log_p

    @property
    def device(self):
        return self.x.device

    def to(self, device):
        self.x = self.x.to(device)
        self.log_q = self.log_q.to(device)
        self.log_p = self.log_p.to(device)
        self.grad_log_q = self.grad_log_q.to(device) if self.grad_log_q is not None else None
        self.grad_log_p = self.grad_log_p.to(device) if self.grad_log_p is not None else None

    def __getitem__(self, indices):
        gq = self.grad_log_q[indices] if self.grad_log_q is not None else None
        gp = self.grad_log_p[indices] if self.grad_log_p is not None else None
        return Point(self.x[indices], self.log_q[indices], self.log_p[indices], gq, gp)

    def __setitem__(self, indices, values):
        self.x[indices] = values.x
        self.log_q[indices] = values.log_q
        self.log_p[indices] = values.log_p
        if self.grad_log_q is not None:
            self.grad_log_q[indices] = values.grad_log_q
            self.grad_log_p[indices] = values.grad_log_p

    def __len__(self):
        return self.x.shape[0]
