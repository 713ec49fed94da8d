"""Multi-GPU AIS: independent chains sharded over ranks (one process per GPU), ONE all-gather of the
final particles / log-weights (RCCL over xGMI through torch.distributed's "nccl" backend).

The reference has no distributed code; chains are independent everywhere in the path
(§8e), so rank r owns chains [r*B/R, (r+1)*B/R) with replicated flow/target parameters and there is no
data-path collective until the particles are gathered.  Payload per rank: [B/R, D+2] fp32 (x | log_w |
log_q) — a few hundred KiB, latency-bound on xGMI, hence a single direct all-gather (no ring of
small buckets).

Metropolis noise scalings (metropolis.py:68-73) adapt on the whole batch too: scaling (i, n) is read by update n of transition i
only, so a sharded call defers the rule of all M transitions to ONE slab gather at its end (`run_metropolis_deferred`).

Step sizes: the reference adapts every transition's step size on the mean acceptance of the WHOLE batch
(hmc.py:122-123,162-170).  `ShardedAnnealedImportanceSampler` keeps exactly that: each transition publishes its acceptance
sums per 16-chain block (a slab of 2 ceil(B/16R) + 1 floats), ONE tiny all-gather per transition joins the slabs in rank
order and every rank applies the rule to the same numbers (fabhip_hmc_adapt_gathered) - with shards that are multiples
of 16 chains the sums are added in the order a single device uses, so the sharded run reproduces the single-device step
sizes bit for bit.  With tuning frozen (`set_eval_mode(True)`, the reference's own evaluation setting) no such collective
is needed and the rank-local call is the fused one.  `ShardedAIS` (per-rank adaptation, optional averaging) is kept
for callers that bring their own rank-local sampler.
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


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def all_gather_rows(buf: torch.Tensor, group=None) -> torch.Tensor:
    """[n, ...] on every rank -> [world * n, ...] in rank order: ONE all-gather.  RCCL ("nccl") takes device tensors
    directly; a host-only backend (gloo: the CPU tests, and the 2-process run on a 1-GPU box) is fed through the host."""
    world = _world(group)
    if world == 1:
        return buf
    out = torch.empty((world * buf.shape[0],) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    if buf.is_cuda and dist.get_backend(group) != "nccl":
        host = torch.empty(out.shape, dtype=buf.dtype)
        dist.all_gather_into_tensor(host, buf.cpu().contiguous(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    return out


def gather_particles(x, log_w, log_q, capacity: int, group=None, compact: bool = True):
    """All-gather fixed-size shards (invalid rows stay in place with log_w = -inf, compaction after).
    compact=False skips the compaction - and with it the host synchronisation of the boolean-mask indexing: the result
    keeps `world * capacity` rows, dropped chains appear as rows with log_w = -inf (weight 0 for ESS / log Z /
    resampling) and x = 0."""
    buf = all_gather_rows(pack_particles(x, log_w, log_q, capacity), group)
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


class HipShardBackend:
    """The rank-local pieces of one sharded AIS call on the GPU (torch.ops.fabhip.ais_phase / hmc_adapt_gathered over
    an `AnnealedImportanceSampler` whose plug-ins are fabhip-native).  `ShardedAnnealedImportanceSampler` drives these
    four methods; the CPU tests drive the same loop with an oracle-backed stand-in."""

    def __init__(self, ais):
        from . import _ops
        from .transition_operators import HamiltonianMonteCarlo
        self.ais, self.ops, self._ops_mod = ais, _ops.load(), _ops
        self.op = ais.transition_operator
        self.hmc = isinstance(self.op, HamiltonianMonteCarlo)

    @property
    def n_transitions(self) -> int:
        return self.ais.n_intermediate_distributions

    @property
    def tuning(self) -> bool:
        """True when a transition's step size depends on the acceptance of the whole batch: HMC outside eval mode (one
        slab gather per transition: the next transition uses the adapted common step size), Metropolis with
        `adjust_step_size` outside eval mode (`run_metropolis_deferred`: ONE gather at the end of the call)."""
        if self.hmc:
            return not self.op.eval_mode
        return bool(getattr(self.op, "adjust_step_size", False)) and not self.op.eval_mode

    def run_metropolis_deferred(self, b, eps0=None, noise_a=None, noise_b=None):
        """This shard's whole Metropolis AIS call (one fabhip_ais_phase: INIT, transitions 1 .. M, FINISH) with the
        noise-scaling rule of metropolis.py:68-73 deferred: scaling (i, n) is read by update n of transition i only, so
        nothing inside the call depends on the adjusted values and the block sums of all M transitions travel in ONE slab
        (fabhip_metropolis_partials_floats).  Returns (Point, log_w, slab); `adapt_metropolis` applies the rule to the
        gathered slabs."""
        from .transition_operators import Metropolis
        op, ais = self.op, self.ais
        if not isinstance(op, Metropolis) or not ais.is_native:
            raise self._ops_mod.FabhipError("exact sharded noise-scaling adaptation needs fab_torch_amd's Metropolis over a "
                                            "RealNVP flow and a native target; other plug-ins: set_eval_mode(True)")
        if bool(op.p_target) != bool(ais.p_target) or (not ais.p_target and op.alpha != ais.alpha):
            raise self._ops_mod.FabhipError("AIS and transition operator disagree on p_target / alpha")
        flow, target = ais._native_parts()
        dev = flow._nf_model.q0.loc.device
        D, M, nu = flow.dim, self.n_transitions, int(op.n_updates)
        f32 = dict(dtype=torch.float32, device=dev)
        counts_stats = torch.zeros(18, **f32)
        st = {"b": int(b), "x": torch.empty((b, D), **f32), "lq": torch.empty(b, **f32), "lp": torch.empty(b, **f32),
              "log_w": torch.empty(b, **f32), "n_valid": counts_stats[16:18].view(torch.int32), "stats": counts_stats[:16]}
        eps0 = torch.randn((b, D), **f32) if eps0 is None else eps0.contiguous()
        noise_a = torch.randn((M, nu, b, D), **f32) if noise_a is None else noise_a.contiguous()
        noise_b = torch.rand((M, nu, b), **f32) if noise_b is None else noise_b.contiguous()
        slab = torch.empty(int(self.ops.metropolis_partials_floats(int(b), M, nu)), **f32)
        alpha = float(ais.alpha) if ais.alpha is not None else 0.0
        self.ops.ais_phase(*flow.native(), *target.native_target(), ais._betas(), alpha, bool(ais.p_target),
                           self._ops_mod.TRANSITION_METROPOLIS, 3, 1, M, eps0, noise_a, noise_b, op.noise_scalings, None, None,
                           nu, 0, 0.0, float(op.target_prob_accept), True, st["x"], st["lq"], st["lp"], None, None,
                           st["log_w"], st["n_valid"], st["stats"], slab, None, None, None, None, None, None,
                           self._ops_mod.precision_of(flow))
        pt, log_w = self._collect(st, grads=False)
        return pt, log_w, slab

    def adapt_metropolis(self, gathered, world, b):
        op = self.op
        self.ops.metropolis_adapt_gathered(gathered, int(world), int(b), op.noise_scalings, float(op.target_prob_accept), True)

    def run_fused(self, b, eps0=None, noise_a=None, noise_b=None):
        from .ais import NoValidPoints
        from .point import Point
        self.empty_phase = None
        try:
            pt, log_w = self.ais.sample_and_log_weights(b, eps0=eps0, noise_a=noise_a, noise_b=noise_b)
        except NoValidPoints as e:                        # the reference's "No valid points ..." (ais.py:201,211) of THIS shard,
            self.empty_phase = e.phase                    # and nothing else (the message text of any Exception was matched)
            # an empty shard - the gathered set decides (see finish).  The shape of an empty Point comes from the sampler's own
            # plug-ins (native flows have `dim`; a generic base distribution its `event_shape`)
            flow = self.ais.base_distribution
            D = int(getattr(flow, "dim", None) or flow.event_shape[0])
            try:
                dev = next(flow.parameters()).device
            except (StopIteration, AttributeError):
                dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)      # noqa: E731
            pt, log_w = Point(z(0, D), z(0), z(0), z(0, D), z(0, D)), z(0)
        return pt, log_w

    def _common(self, st):
        ais, op = self.ais, self.op
        flow, target = ais._native_parts()
        alpha = float(ais.alpha) if ais.alpha is not None else 0.0
        return (*flow.native(), *target.native_target(), ais._betas(), alpha, bool(ais.p_target),
                self._ops_mod.TRANSITION_HMC)

    def _phase(self, st, phases, j0, j1, partials=None, tune=False):
        op = self.op
        self.ops.ais_phase(*self._common(st), int(phases), int(j0), int(j1), st["eps0"], st["noise_a"], st["noise_b"],
                           op.epsilons, op.common_epsilon, op.mass_vector, 1, op.L, float(op.max_grad),
                           float(op.target_p_accept), bool(tune), st["x"], st["lq"], st["lp"], st["gq"], st["gp"],
                           st["log_w"], st["n_valid"], st["stats"], partials, None, None, None, None, None, None,
                           self._ops_mod.precision_of(self.ais.base_distribution))

    def one_op_available(self, group=None) -> bool:
        """Can `ais_sharded_tuned` find this process group from C++?  (It resolves the group by NAME in c10d's registry;
        torch registers every group it creates.)  Checked once per sampler, BEFORE anything is enqueued and from state that
        is the same on every rank, so that all ranks take the same form of the loop."""
        try:
            pg = group if group is not None else dist.distributed_c10d._get_default_group()
            import torch._C._distributed_c10d as c10d
            return hasattr(self.ops, "ais_sharded_tuned") and c10d._resolve_process_group(str(pg.group_name)) is not None
        except Exception:                                  # noqa: BLE001 - any failure: the Python-stepped loop
            return False

    def run_tuned(self, b, group=None, eps0=None, noise_a=None, noise_b=None):
        """The whole tuned call of this shard in ONE op (torch.ops.fabhip.ais_sharded_tuned, csrc/torch_ops.cpp): the loop
        `begin` / M x (`step`, slab all-gather, `adapt`) / `finish` with the collectives issued from C++ through the c10d
        process group (on RCCL: ordered on the compute stream, the host never blocks).  Returns (Point, log_w, collectives)."""
        st = self._state(b, eps0, noise_a, noise_b)
        op = self.op
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        n = self.ops.ais_sharded_tuned(*self._common(st), st["eps0"], st["noise_a"], st["noise_b"], op.epsilons,
                                       op.common_epsilon, op.mass_vector, op.L, float(op.max_grad),
                                       float(op.target_p_accept), st["x"], st["lq"], st["lp"], st["gq"], st["gp"],
                                       st["log_w"], st["n_valid"], st["stats"], st["slab"], op._p_accept_first,
                                       op._p_accept_last, op._dist_first, op._dist_last,
                                       self._ops_mod.precision_of(self.ais.base_distribution), str(pg.group_name))
        pt, log_w = self._collect(st)
        return pt, log_w, int(n)

    def begin(self, b, eps0=None, noise_a=None, noise_b=None):
        """Chain initialisation + "chain init" filter + base ESS of this rank's b chains (FABHIP_AIS_INIT)."""
        st = self._state(b, eps0, noise_a, noise_b)
        self._phase(st, 1, 1, 0)
        return st

    def _state(self, b, eps0=None, noise_a=None, noise_b=None):
        """Checks + the state tensors of one tuned call (noise drawn here: same generator, same order in both forms)."""
        if not self.hmc or self.op.n_outer != 1:
            raise self._ops_mod.FabhipError("exact sharded step-size adaptation: HMC with n_outer == 1 "
                                            "(every shipped config, experiments/setup_run.py:190)")
        if not self.ais.is_native:
            raise self._ops_mod.FabhipError("sharded AIS with step-size tuning on needs a RealNVP flow and a native target "
                                            "(fabhip_ais_phase); other plug-ins: set_eval_mode(True), or tune on one rank")
        op, ais = self.op, self.ais
        if bool(op.p_target) != bool(ais.p_target) or (not ais.p_target and op.alpha != ais.alpha):
            raise self._ops_mod.FabhipError("AIS and transition operator disagree on p_target / alpha")
        flow, _ = self.ais._native_parts()
        dev = flow._nf_model.q0.loc.device
        D, M = flow.dim, self.n_transitions
        f32 = dict(dtype=torch.float32, device=dev)
        counts_stats = torch.zeros(18, **f32)                 # stats[16] | n_valid[2]: ONE device->host read at the end
        st = {"b": int(b),
              "eps0": (torch.randn((b, D), **f32) if eps0 is None else eps0.contiguous()),
              "noise_a": (torch.randn((M, 1, b, D), **f32) if noise_a is None else noise_a.contiguous()),
              "noise_b": (torch.empty((M, 1, b), **f32).exponential_(1.0) if noise_b is None else noise_b.contiguous()),
              "x": torch.empty((b, D), **f32), "lq": torch.empty(b, **f32), "lp": torch.empty(b, **f32),
              "gq": torch.empty((b, D), **f32), "gp": torch.empty((b, D), **f32), "log_w": torch.empty(b, **f32),
              "n_valid": counts_stats[16:18].view(torch.int32), "stats": counts_stats[:16],
              "slab": torch.empty(int(self.ops.hmc_partials_floats(int(b))), **f32)}
        return st

    def step(self, st, j) -> torch.Tensor:
        """Transition j with the adaptation deferred: returns this rank's acceptance slab."""
        self._phase(st, 0, j, j, partials=st["slab"], tune=True)
        return st["slab"]

    def adapt(self, st, j, gathered, world):
        op, M = self.op, self.n_transitions
        pa, ad = (op._p_accept_first, op._dist_first) if j == 1 else ((op._p_accept_last, op._dist_last) if j == M
                                                                      else (None, None))
        self.ops.hmc_adapt_gathered(gathered, int(world), st["b"], op.epsilons[j - 1], op.common_epsilon,
                                    float(op.target_p_accept), True, pa, ad)

    def finish(self, st):
        """"chain end" filter + local ESS / log Z (FABHIP_AIS_FINISH); one device->host read for the row counts."""
        self._phase(st, 2, 1, 0)
        return self._collect(st)

    def _collect(self, st, grads=True):
        from .point import Point
        hs, (n_init, n_end) = self._ops_mod.read_counts_and_stats(st["n_valid"], st["stats"])
        # a shard without survivors is NOT an error here: the other ranks are about to enter the particle all-gather, and one
        # device holding every chain would only fail if NO chain survived - the caller decides from the gathered set, on every
        # rank alike (a rank-local raise left the others blocked in the collective)
        from .ais import LoggingInfo                      # this rank's own chains (the gathered set: logging_info)
        self.ais._logging_info = LoggingInfo(ess_base=float(hs[0]), ess_ais=float(hs[3]), log_Z=float(hs[4]))
        pt = Point(st["x"][:n_end], st["lq"][:n_end], st["lp"][:n_end], st["gq"][:n_end] if grads else None,
                   st["gp"][:n_end] if grads else None)
        return pt, st["log_w"][:n_end].detach()


class ShardedAnnealedImportanceSampler:
    """`AnnealedImportanceSampler.sample_and_log_weights(total_batch)` over the ranks of a process group with the
    semantics of ONE device holding every chain: rank r runs chains [r b, (r + 1) b), the step sizes adapt on the
    acceptance of all chains (one slab all-gather per transition while tuning is on), the particles are joined by one
    all-gather at the end, ESS / log Z are those of the gathered set.  Returns (x, log_w, log_q) of all chains on every
    rank (`compact=False`: fixed `world * max shard` rows, dropped chains - and the padding of an uneven split - as log_w = -inf rows; the fused rank-local call and the
    "chain end" filter each read their row counts once, the gather itself adds no host synchronisation).  A shard that loses all
    of its chains is an empty shard, not an error: "No valid points" is raised - on every rank - only when the GATHERED set is
    empty (compact=True; with compact=False `logging_info` shows it).  Bit-for-bit equality with one device holds when both
    use the same tile shape (`FABHIP_OPT_TILE_SHAPE`, or batches that select the same one: chains per workgroup follow the
    LOCAL batch)."""

    def __init__(self, ais=None, group=None, backend=None, one_op: Optional[bool] = None):
        import os
        self.group = group
        self.backend = backend if backend is not None else HipShardBackend(ais)
        # tuned calls: one op per call (default) or the Python-stepped loop it replaced (FABHIP_SHARDED_ONE_OP=0: kept as
        # the reference implementation - tests/test_gpu_sharded.py compares the two bit for bit)
        self.one_op = (os.environ.get("FABHIP_SHARDED_ONE_OP", "1") != "0") if one_op is None else bool(one_op)
        self._one_op_ok = None
        self.logging_info = None
        self.n_slab_gathers = 0                 # collectives issued by the last call besides the particle gather

    def _use_one_op(self) -> bool:
        if not (self.one_op and hasattr(self.backend, "run_tuned")):
            return False
        if self._one_op_ok is None:                        # once: the same answer on every rank (see one_op_available)
            chk = getattr(self.backend, "one_op_available", None)
            self._one_op_ok = bool(chk(self.group)) if chk is not None else True
            if not self._one_op_ok:
                import warnings
                warnings.warn("fab_torch_amd: the process group cannot be resolved from C++ - tuned sharded AIS falls back to "
                              "the Python-stepped loop (same values)")
        return self._one_op_ok

    def local_batch(self, total_batch: int) -> int:
        """This rank's share of `total_batch` chains.  Even splits everywhere; an UNEVEN split (the first total % world ranks run
        one chain more, `shard_sizes`) is taken where no slab travels - tuning frozen / one rank: the particle gather pads every
        shard to the largest one with invalid rows.  With tuning on the slabs of all ranks must have one shape (and only shards
        that are multiples of 16 chains reproduce one device bit for bit): refused."""
        world = _world(self.group)
        if total_batch % world and world > 1 and self.backend.tuning:
            raise ValueError(f"sharded AIS with step-size tuning on: {total_batch} chains do not split evenly over {world} ranks "
                             "(the acceptance slabs of all ranks must have one shape); set_eval_mode(True) takes uneven splits")
        return shard_sizes(total_batch, world)[_rank(self.group)]

    def sample_and_log_weights(self, total_batch: int, eps0=None, noise_a=None, noise_b=None, compact: bool = True,
                               logging: bool = True):
        world = _world(self.group)
        b = self.local_batch(total_batch)
        be = self.backend
        self.n_slab_gathers = 0
        if world == 1 or not be.tuning:
            pt, log_w = be.run_fused(b, eps0, noise_a, noise_b)
        elif not getattr(be, "hmc", True):                 # Metropolis: the whole call, then ONE slab gather + the rule
            pt, log_w, slab = be.run_metropolis_deferred(b, eps0, noise_a, noise_b)
            be.adapt_metropolis(all_gather_rows(slab.reshape(1, -1), self.group).reshape(-1), world, b)
            self.n_slab_gathers = 1
        elif self._use_one_op():                           # the loop below, inside one op (collectives issued from C++)
            pt, log_w, self.n_slab_gathers = be.run_tuned(b, self.group, eps0, noise_a, noise_b)
        else:
            st = be.begin(b, eps0, noise_a, noise_b)
            for j in range(1, be.n_transitions + 1):
                slab = be.step(st, j)
                gathered = all_gather_rows(slab.reshape(1, -1), self.group).reshape(-1)
                self.n_slab_gathers += 1
                be.adapt(st, j, gathered, world)
            pt, log_w = be.finish(st)
        cap = max(shard_sizes(total_batch, world))         # (== b for an even split)
        buf = all_gather_rows(pack_particles(pt.x, log_w, pt.log_q, cap), self.group)
        D = buf.shape[-1] - 3
        n_valid = buf[:, D + 2].sum()                      # device scalar: chains that survived on any rank
        if compact:                                        # (the boolean-mask indexing synchronises with the host anyway)
            x, lw, lq = unpack_particles(buf)
            if x.shape[0] == 0:                            # every rank sees the same gathered set: all raise together
                from .ais import NoValidPoints
                # (one rank: the phase in which this rank's chains died, as the reference reports it; several ranks may have lost
                #  theirs in different phases - "end" then: the gathered set is what is empty)
                phase = getattr(be, "empty_phase", None) if world == 1 else None
                raise NoValidPoints(phase or "end")
        else:
            x, lw, lq = buf[:, :D], buf[:, D], buf[:, D + 1]
        if logging:
            self.logging_info = self._global_stats(lw, total_batch, n_valid)
        return x, lw, lq

    @staticmethod
    def _global_stats(log_w, total_batch, n_valid=None):
        """ESS and log Z of the gathered set (ais.py:80-86: ESS = (sum w)^2 / sum w^2 / N with N the chains that SURVIVED -
        dropped chains are padding rows with weight 0 here -, log Z normalised by the REQUESTED batch), lazily: device scalars,
        no synchronisation.  On the GPU: fabhip_ess_logz, the kernel the single-device call uses."""
        n = log_w.shape[0]
        nv = torch.isfinite(log_w).sum() if n_valid is None else n_valid
        if log_w.is_cuda:
            from . import _ops
            out = _ops.load().ess_logz(log_w.detach().contiguous().float(), None, float(total_batch))   # ESS / n rows, log Z
            return {"ess_ais": out[0].double() * (float(n) / nv.double()), "log_Z": out[1].double()}
        lw = log_w.double()
        lse = torch.logsumexp(lw, 0)
        ess = torch.exp(2 * lse - torch.logsumexp(2 * lw, 0)) / nv
        return {"ess_ais": ess, "log_Z": lse - torch.log(torch.tensor(float(total_batch), dtype=torch.float64))}
