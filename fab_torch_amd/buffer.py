"""Prioritised replay buffer on the device — same interface and semantics as
fab/utils/prioritised_replay_buffer.py:20-153: ring buffer of (x, log_w, log_q_old); sampling without
replacement by the Gumbel-top-k trick over log_w followed by a random permutation (:10-17); `adjust` adds
the log-weight correction, refreshes log_q_old and kills entries whose correction is not finite (log_w = -inf,
:117-131).  All tensors stay on the buffer device (the GPU): no host round trip per iteration.
("Next" row of SURVEY.md section 8f: device-side PyTorch ops for now, a radix-select top-k kernel later.)"""
from typing import Callable, Iterable, NamedTuple, Tuple

import torch

from . import _ops


class ReplayData(NamedTuple):
    x: torch.Tensor
    log_w: torch.Tensor
    log_q_old: torch.Tensor


TOPK_SORT_MAX = 16384     # sorted mode of fabhip_topk sorts the selection in one workgroup's LDS


def topk_indices(keys: torch.Tensor, k: int, sorted: bool = False) -> torch.Tensor:
    """Indices of the k largest keys — fabhip_topk (radix select + index-ordered compaction).  sorted=False: in
    ascending index order; sorted=True (k <= 16384): descending key order, ties by ascending index."""
    _ops.require_device(keys, "keys")
    return _ops.load().topk(keys.detach().contiguous().float(), int(k), bool(sorted))


def sample_without_replacement(logits: torch.Tensor, n: int, gumbel: torch.Tensor = None,
                               perm: torch.Tensor = None) -> torch.Tensor:
    """Gumbel-max trick: top-n of logits + Gumbel(0,1) noise, in random order
    (fab/utils/prioritised_replay_buffer.py:10-17).  `gumbel [len(logits)]` / `perm [n]` may be supplied (explicit
    noise, as everywhere in this package: parity replays of the reference's draws); otherwise they come from the
    device generator."""
    if gumbel is None and perm is None and logits.is_cuda and logits.dtype == torch.float32:
        # both draws, the Gumbel keys, the selection and its random order in ONE op (fabhip::buffer_sample_indices): the
        # expressions below launch ~45 small kernels from Python
        return _ops.load().buffer_sample_indices(logits.detach().contiguous(), int(n))
    if gumbel is None:
        u = torch.rand(logits.shape, device=logits.device, dtype=logits.dtype).clamp_(min=torch.finfo(logits.dtype).tiny)
        gumbel = -torch.log(-torch.log(u))
    indices = topk_indices(gumbel.to(logits.device) + logits, n)    # a set, in index order (GPU only): permuted below
    if perm is None:
        return indices[torch.randperm(n, device=indices.device)]
    # the reference permutes torch.topk(sorted=False)'s output, whose order is unspecified; a replay therefore has
    # to present the SAME set in the reference's pre-permutation order, which the caller encodes in `perm` as
    # positions into the ascending-index order used here (see tests/test_gpu_workloads.py)
    return indices[perm.to(indices.device)]


class PrioritisedReplayBuffer:
    def __init__(self, dim: int, max_length: int, min_sample_length: int,
                 initial_sampler: Callable[[], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], device: str = "cpu",
                 sample_with_replacement: bool = False, fill_buffer_during_init: bool = True):
        assert min_sample_length < max_length
        # the reference defaults to a host buffer and samples it with torch.topk / Categorical on the CPU; here sample()
        # always runs the HIP top-k / multinomial kernels.  add() / adjust() are plain tensor code and work anywhere (the
        # CPU tests pin them against the reference's buffer), so a host buffer is allowed but announced up front
        if torch.device(device).type != "cuda":
            import warnings
            warnings.warn(f"PrioritisedReplayBuffer(device={device!r}): sample() needs the buffer on the GPU (there is no "
                          "CPU sampling path in this package) - pass device='cuda' / the flow's device", stacklevel=2)
        self.dim, self.max_length, self.min_sample_length = dim, max_length, min_sample_length
        self.buffer = ReplayData(x=torch.zeros(max_length, dim, device=device),
                                 log_w=torch.zeros(max_length, device=device),
                                 log_q_old=torch.zeros(max_length, device=device))
        self.device = device
        self.current_index = 0
        self.is_full = False
        self.can_sample = False
        self.sample_with_replacement = sample_with_replacement
        if fill_buffer_during_init:
            while not self.can_sample:
                self.add(*initial_sampler())
        else:
            print("Buffer not initialised, expected that checkpoint will be loaded.")

    @torch.no_grad()
    def add(self, x: torch.Tensor, log_w: torch.Tensor, log_q_old: torch.Tensor) -> None:
        n = x.shape[0]
        if self.buffer.x.is_cuda and n <= self.max_length:
            dev = self.buffer.x.device                     # one launch (fabhip::buffer_add) instead of arange / % / three index_puts
            _ops.load().buffer_add(x.detach().to(dev).float().contiguous(), log_w.detach().to(dev).float().contiguous(),
                                   log_q_old.detach().to(dev).float().contiguous(), int(self.current_index), self.buffer.x,
                                   self.buffer.log_w, self.buffer.log_q_old)
        else:
            idx = (torch.arange(n, device=self.device) + self.current_index) % self.max_length
            self.buffer.x[idx] = x.to(self.device)
            self.buffer.log_w[idx] = log_w.to(self.device)
            self.buffer.log_q_old[idx] = log_q_old.to(self.device)
        new_index = self.current_index + n
        if not self.is_full:
            self.is_full = new_index >= self.max_length
            self.can_sample = new_index >= self.min_sample_length
        self.current_index = new_index % self.max_length

    @torch.no_grad()
    def sample(self, batch_size: int, gumbel: torch.Tensor = None, perm: torch.Tensor = None):
        if not self.can_sample:
            raise Exception("Buffer must be at minimum length before calling sample")
        max_index = self.max_length if self.is_full else self.current_index
        if self.sample_with_replacement:
            from .resample import multinomial_indices         # Categorical(logits=log_w).sample_n (:95-97) on the GPU
            indices = multinomial_indices(self.buffer.log_w[:max_index], batch_size)
        else:
            indices = sample_without_replacement(self.buffer.log_w[:max_index], batch_size, gumbel, perm)
        return self.buffer.x[indices], self.buffer.log_w[indices], self.buffer.log_q_old[indices], indices

    @torch.no_grad()
    def sample_indices(self, batch_size: int, gumbel: torch.Tensor = None, perm: torch.Tensor = None) -> torch.Tensor:
        """The row indices `sample(batch_size)` would gather (:88-97), without gathering: the fused minibatch step
        (`fabhip::buffer_train_step`) reads x and log_q_old in place."""
        if not self.can_sample:
            raise Exception("Buffer must be at minimum length before calling sample")
        max_index = self.max_length if self.is_full else self.current_index
        if self.sample_with_replacement:
            from .resample import multinomial_indices
            return multinomial_indices(self.buffer.log_w[:max_index], batch_size)
        return sample_without_replacement(self.buffer.log_w[:max_index], batch_size, gumbel, perm)

    def sample_n_batches(self, batch_size: int, n_batches: int, gumbel: torch.Tensor = None,
                         perm: torch.Tensor = None) -> Iterable[Tuple[torch.Tensor, ...]]:
        x, log_w, log_q_old, indices = self.sample(batch_size * n_batches, gumbel, perm)
        return list(zip(torch.chunk(x, n_batches), torch.chunk(log_w, n_batches), torch.chunk(log_q_old, n_batches),
                        torch.chunk(indices, n_batches)))

    @torch.no_grad()
    def adjust(self, log_w_adjustment, log_q, indices):
        # same result as the reference's three masked index_puts (:117-131), written without boolean-mask
        # indexing so that no host synchronisation happens inside the minibatch loop
        indices = indices.to(self.device)
        adj, log_q = log_w_adjustment.to(self.device), log_q.to(self.device)
        valid = torch.isfinite(adj) & torch.isfinite(log_q)
        neg_inf = torch.full_like(adj, -float("inf"))
        self.buffer.log_w[indices] = torch.where(valid, self.buffer.log_w[indices] + adj, neg_inf)
        self.buffer.log_q_old[indices] = torch.where(valid, log_q, self.buffer.log_q_old[indices])

    def save(self, path):
        torch.save({'x': self.buffer.x.detach().cpu(), 'log_w': self.buffer.log_w.detach().cpu(),
                    'log_q_old': self.buffer.log_q_old.detach().cpu(), 'current_index': self.current_index,
                    'is_full': self.is_full, 'can_sample': self.can_sample}, path)

    def load(self, path):
        old = torch.load(path)
        self.buffer.x.copy_(old['x'])
        self.buffer.log_w.copy_(old['log_w'])
        self.buffer.log_q_old.copy_(old['log_q_old'])
        self.current_index, self.is_full, self.can_sample = old['current_index'], old['is_full'], old['can_sample']
