"""Multi-GPU AIS: independent chains sharded over ranks (one process per GPU), ONE all-gather of the
final particles / log-weights (RCCL over xGMI through torch.distributed's "nccl" backend).

The reference has no distributed code; chains are independent everywhere in the path
(§8e), so rank r owns chains [r*B/R, (r+1)*B/R) with replicated flow/target parameters and there is no
data-path collective until the particles are gathered.  Payload per rank: [B/R, D+2] fp32 (x | log_w |
log_q) — a few hundred KiB, latency-bound on xGMI, hence a single direct all-gather (no ring of
small buckets).  Step sizes adapt per rank on the local shard (`sync_step_size=False`, default) or are
kept identical across ranks by averaging the adapted values with one tiny all-reduce per call.
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_sizes(total: int, world: int):
    base, rem = divmod(total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def pack_particles(x: torch.Tensor, log_w: torch.Tensor, log_q: torch.Tensor, capacity: int) -> torch.Tensor:
    """[capacity, D+3]: x | log_w | log_q | valid-flag, rows beyond len(x) are padding (flag 0)."""
    n, D = x.shape
    if n == capacity:                           # the usual case (no chain dropped): one concatenation kernel
        lw = log_w.to(torch.float32)
        return torch.cat((x.to(torch.float32), lw[:, None], log_q.to(torch.float32)[:, None],
                          torch.ones_like(lw)[:, None]), dim=1)
    buf = torch.zeros((capacity, D + 3), dtype=torch.float32, device=x.device)
    buf[:n, :D] = x
    buf[:n, D] = log_w
    buf[:n, D + 1] = log_q
    buf[:n, D + 2] = 1.0
    buf[n:, D] = -float("inf")
    return buf


def unpack_particles(buf: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    D = buf.shape[-1] - 3
    flat = buf.reshape(-1, D + 3)
    keep = flat[:, D + 2] > 0.5
    flat = flat[keep]
    return flat[:, :D], flat[:, D], flat[:, D + 1]


def gather_particles(x, log_w, log_q, capacity: int, group=None, compact: bool = True):
    """All-gather fixed-size shards (invalid rows stay in place with log_w = -inf, compaction after).
    compact=False skips the compaction - and with it the host synchronisation of the boolean-mask indexing: the result
    keeps `world * capacity` rows, dropped chains appear as rows with log_w = -inf (weight 0 for ESS / log Z /
    resampling) and x = 0."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    buf = pack_particles(x, log_w, log_q, capacity)
    if world > 1:
        out = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype, device=buf.device)
        dist.all_gather_into_tensor(out, buf, group=group)
        buf = out
    if compact:
        return unpack_particles(buf)
    D = buf.shape[-1] - 3
    return buf[:, :D], buf[:, D], buf[:, D + 1]


class ShardedAIS:
    """`sample_and_log_weights(total_batch)` over all ranks.

    `local_sampler(batch) -> (x, log_w, log_q)` runs the rank-local chains (on a GPU box:
    `AnnealedImportanceSampler.sample_and_log_weights`; in the CPU/gloo tests: the oracle)."""

    def __init__(self, local_sampler: Callable, step_state: Optional[Callable] = None, sync_step_size: bool = False,
                 group=None):
        self.local_sampler = local_sampler
        self.step_state = step_state          # () -> list of tensors holding adapted step sizes
        self.sync_step_size = sync_step_size
        self.group = group

    def sample_and_log_weights(self, total_batch: int):
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        sizes = shard_sizes(total_batch, world)
        x, log_w, log_q = self.local_sampler(sizes[rank])
        if self.sync_step_size and self.step_state is not None and world > 1:
            # ONE tiny all-reduce for all step-size tensors (epsilons [M, n_outer] + common_epsilon [1]: latency-bound)
            ts = list(self.step_state())
            flat = torch.cat([t.reshape(-1).to(torch.float32) for t in ts])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(world)
            o = 0
            for t in ts:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()
        return gather_particles(x, log_w, log_q, max(sizes), self.group)
