"""N>1 path on CPU: world_size-2 `gloo` run of the sharded-AIS host logic (chains sharded over ranks,
one all-gather of fixed-size particle slabs).  The rank-local sampler is the CPU oracle here; on the GPU
box it is the HIP AnnealedImportanceSampler."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT

from fab_torch_amd import parallel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _local_sampler_factory(rank):
    sys.path.insert(0, ROOT)
    from oracle import ais as oais, flow as oflow, targets as otgt
    D, K, M = 6, 2, 3
    torch.manual_seed(0)                                   # replicated parameters
    nf = oflow.make_realnvp(D, K, 5)
    oflow.randomize_last_layers(nf, 0.05, 1)
    target = otgt.ManyWell(D)
    hmc = oais.HMC(M, D, nf.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.15, eval_mode=True)
    ais = oais.AIS(lambda e: tuple(t.detach() for t in nf.sample_eps(e)), nf.log_prob, target.log_prob, hmc, False, 2.0, M)

    def sampler(b):
        g = torch.Generator().manual_seed(100 + rank)      # per-rank noise stream
        eps0 = torch.randn(b, D, generator=g)
        if rank == 1 and b > 2:
            eps0[1, 0] = float("nan")                      # one invalid chain on rank 1 -> filtered locally
        noise_p = torch.randn(M, 1, b, D, generator=g)
        noise_e = torch.empty(M, 1, b).exponential_(generator=g)
        pt, lw, _ = ais.sample_and_log_weights(eps0, noise_p, noise_e)
        return pt.x, lw, pt.log_q
    return sampler


def _worker(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sh = parallel.ShardedAIS(lambda b: _local_sampler_factory(rank)(b))
    x, lw, lq = sh.sample_and_log_weights(total)
    if rank == 0:
        torch.save({"x": x, "lw": lw, "lq": lq}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_sizes_and_packing_roundtrip():
    assert parallel.shard_sizes(10, 4) == [3, 3, 2, 2] and sum(parallel.shard_sizes(16384, 8)) == 16384
    x, lw, lq = torch.randn(5, 3), torch.randn(5), torch.randn(5)
    buf = parallel.pack_particles(x, lw, lq, 8)
    assert buf.shape == (8, 6) and torch.isinf(buf[5:, 3]).all()
    x2, lw2, lq2 = parallel.unpack_particles(buf)
    assert torch.equal(x2, x) and torch.equal(lw2, lw) and torch.equal(lq2, lq)


def test_two_rank_gloo_gather_equals_concatenated_shards(tmp_path):
    world, total = 2, 21                                   # uneven shards: 11 + 10
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = torch.load(out)
    xs, lws = [], []
    for r, b in enumerate(parallel.shard_sizes(total, world)):
        x, lw, lq = _local_sampler_factory(r)(b)
        xs.append(x); lws.append(lw)
    x_ref, lw_ref = torch.cat(xs), torch.cat(lws)
    assert got["x"].shape[0] == total - 1                 # the NaN chain of rank 1 was removed, no padding leaked
    np.testing.assert_array_equal(got["x"].numpy(), x_ref.numpy())
    np.testing.assert_array_equal(got["lw"].numpy(), lw_ref.numpy())


def _worker_sync(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eps = torch.full((3, 1), 0.1 * (rank + 1))             # per-rank adapted step sizes: 0.1 / 0.2
    ceps = torch.tensor([0.01 * (rank + 1)])

    def sampler(b):
        return torch.zeros(b, 2), torch.zeros(b), torch.zeros(b)
    sh = parallel.ShardedAIS(sampler, step_state=lambda: [eps, ceps], sync_step_size=True)
    sh.sample_and_log_weights(8)
    torch.save({"eps": eps, "ceps": ceps}, out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_size_averaging_is_one_collective_and_identical_on_all_ranks(tmp_path):
    out = str(tmp_path / "s")
    mp.spawn(_worker_sync, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + "0"), torch.load(out + "1")
    assert torch.equal(a["eps"], b["eps"]) and torch.equal(a["ceps"], b["ceps"])
    assert torch.allclose(a["eps"], torch.full((3, 1), 0.15)) and torch.allclose(a["ceps"], torch.tensor([0.015]))


def _worker_nocompact(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sizes = parallel.shard_sizes(total, world)
    x, lw, lq = _local_sampler_factory(rank)(sizes[rank])
    xg, lwg, lqg = parallel.gather_particles(x, lw, lq, max(sizes), compact=False)
    if rank == 0:
        torch.save({"x": xg.clone(), "lw": lwg.clone(), "lq": lqg.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_without_compaction_keeps_fixed_size_and_marks_dropped_chains(tmp_path):
    """compact=False (what bench.py times on N > 1: no boolean-mask indexing, no host synchronisation): world * capacity
    rows, dropped / padding rows carry log_w = -inf and x = 0, and the finite rows are exactly the compacted result."""
    world, total = 2, 21
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_nocompact, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = torch.load(out)
    cap = max(parallel.shard_sizes(total, world))
    assert got["x"].shape[0] == world * cap and got["lw"].shape[0] == world * cap
    keep = torch.isfinite(got["lw"])
    assert int((~keep).sum()) == world * cap - (total - 1)          # one NaN chain on rank 1 + one padding row
    assert bool((got["x"][~keep] == 0).all())
    xs, lws = [], []
    for r, b in enumerate(parallel.shard_sizes(total, world)):
        x, lw, lq = _local_sampler_factory(r)(b)
        xs.append(x); lws.append(lw)
    np.testing.assert_array_equal(got["x"][keep].numpy(), torch.cat(xs).numpy())
    np.testing.assert_array_equal(got["lw"][keep].numpy(), torch.cat(lws).numpy())


# ---- exact cross-rank step-size adaptation (hmc.py:122-123,162-170: the rule sees the acceptance of ALL chains) ------
class _OracleShardBackend:
    """Stand-in for parallel.HipShardBackend on the CPU: the oracle's HMC with the adaptation taken out and restated the
    way the kernels do it (per-16-chain-block fp32 sums of min(1, acceptance), then one sequential sum over the slabs
    in rank order)."""
    tuning = True

    def __init__(self):
        sys.path.insert(0, ROOT)
        from oracle import ais as oais, flow as oflow, targets as otgt
        self.oais = oais
        self.D, self.M = 6, 4
        torch.manual_seed(0)
        self.nf = oflow.make_realnvp(self.D, 2, 5)
        oflow.randomize_last_layers(self.nf, 0.05, 1)
        self.target = otgt.ManyWell(self.D)
        self.hmc = oais.HMC(self.M, self.D, self.nf.log_prob, self.target.log_prob, alpha=2.0, p_target=False, epsilon=0.25,
                            L=3, eval_mode=True)          # (eval_mode: its own adaptation off - `adapt` below does it)
        self.betas = oais.beta_schedule(self.M)
        self.trace = []

    n_transitions = property(lambda self: self.M)

    def begin(self, b, eps0, noise_a, noise_b):
        oais = self.oais
        x, lq0 = (t.detach() for t in self.nf.sample_eps(eps0))
        pt = oais.create_point(x, self.nf.log_prob, self.target.log_prob, with_grad=True)
        lw = (oais.intermediate_log_prob(pt, self.betas[1], 2.0, False) - lq0).detach()
        return {"b": b, "pt": pt, "lw": lw, "na": noise_a, "nb": noise_b}

    def step(self, st, j):
        oais, b = self.oais, st["b"]
        st["pt"] = self.hmc.transition(st["pt"], j, self.betas[j], st["na"][j - 1], st["nb"][j - 1])
        num = oais.intermediate_log_prob(st["pt"], self.betas[j + 1], 2.0, False)
        den = oais.intermediate_log_prob(st["pt"], self.betas[j], 2.0, False)
        st["lw"] = st["lw"] + (num - den)
        log_acc = self.hmc.last_margin - st["nb"][j - 1][0, :b]
        log_acc = torch.nan_to_num(log_acc, nan=-float("inf"), posinf=-float("inf"), neginf=-float("inf"))
        contrib = torch.exp(torch.clamp(log_acc, max=0.0)).float()
        nblk = (b + 15) // 16
        acc = torch.zeros(nblk)
        for k in range(nblk):
            s = torch.tensor(0.0)
            for v in contrib[16 * k:16 * k + 16]:
                s = s + v                                 # fp32, row order (k_hmc_step's block partial)
            acc[k] = s
        return torch.cat([acc, torch.zeros(nblk), torch.tensor([float(b)])])

    def adapt(self, st, j, gathered, world):
        nblk = (st["b"] + 15) // 16
        slabs = gathered.reshape(world, 2 * nblk + 1)
        s, nv = torch.tensor(0.0), 0
        for r in range(world):                            # k_hmc_adapt_gathered: ranks, then blocks, in order
            for k in range(nblk):
                s = s + slabs[r, k]
            nv += int(slabs[r, 2 * nblk])
        up = bool(torch.log(s) - torch.log(torch.tensor(float(nv))) > torch.log(torch.tensor(0.65)))
        h = self.hmc
        h.epsilons[j - 1, 0] = h.epsilons[j - 1, 0] * 1.05 if up else h.epsilons[j - 1, 0] / 1.05
        h.common_epsilon = h.common_epsilon * 1.02 if up else h.common_epsilon / 1.02
        self.trace.append((up, float(s)))

    def finish(self, st):
        return st["pt"], st["lw"]


def _sharded_noise(total, D, M):
    g = torch.Generator().manual_seed(11)
    eps0 = torch.randn(total, D, generator=g)
    na = torch.randn(M, 1, total, D, generator=g)
    na[:, :, total // 2:] *= 2.0                          # the second shard's chains accept far less often than the first's
    nb = torch.empty(M, 1, total).exponential_(generator=g)
    return eps0, na, nb


def _worker_exact(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    be = _OracleShardBackend()
    sh = parallel.ShardedAnnealedImportanceSampler(backend=be)
    b = total // world
    eps0, na, nb = _sharded_noise(total, be.D, be.M)
    sl = slice(rank * b, (rank + 1) * b)
    x, lw, lq = sh.sample_and_log_weights(total, eps0=eps0[sl], noise_a=na[:, :, sl].contiguous(),
                                          noise_b=nb[:, :, sl].contiguous())
    torch.save({"x": x, "lw": lw, "eps": be.hmc.epsilons.clone(), "ceps": be.hmc.common_epsilon.clone(),
                "trace": be.trace, "n_gathers": sh.n_slab_gathers,
                "ess": float(sh.logging_info["ess_ais"]), "log_Z": float(sh.logging_info["log_Z"])}, out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_run_reproduces_the_single_device_step_size_trajectory(tmp_path):
    """SURVEY 8e / VERDICT r2 #5: with tuning ON the step-size rule must see the acceptance of the whole batch.  Two gloo
    ranks x 32 chains against ONE process holding all 64 chains (same noise rows): identical up / down decisions, step
    sizes bit for bit on both ranks, exactly one slab all-gather per transition - and the shards are built so that a
    per-rank rule would have decided differently."""
    world, total = 2, 64
    out = str(tmp_path / "e")
    mp.spawn(_worker_exact, args=(world, _free_port(), total, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    # the single-device run: same pieces, all chains in one process, its own slab as the "gathered" set
    torch.set_num_threads(1)
    be = _OracleShardBackend()
    eps0, na, nb = _sharded_noise(total, be.D, be.M)
    st = be.begin(total, eps0, na, nb)
    local_decisions = []
    for j in range(1, be.M + 1):
        slab = be.step(st, j)
        nblk = total // 16
        for half in (slice(0, nblk // 2), slice(nblk // 2, nblk)):           # what each rank ALONE would have decided
            s = float(slab[half].sum())
            local_decisions.append(s / (total // 2) > 0.65)
        be.adapt(st, j, slab, 1)
    pt, lw = be.finish(st)
    for r in (r0, r1):
        assert torch.equal(r["eps"], be.hmc.epsilons) and torch.equal(r["ceps"], be.hmc.common_epsilon)
        assert [t[0] for t in r["trace"]] == [t[0] for t in be.trace]
        assert [t[1] for t in r["trace"]] == [t[1] for t in be.trace]          # the very same fp32 sums
        assert r["n_gathers"] == be.M
    per_rank = list(zip(local_decisions[0::2], local_decisions[1::2]))
    assert any(a != b for a, b in per_rank), "test set-up: the two shards must disagree on some transition"
    assert torch.equal(r0["x"], r1["x"]) and torch.equal(r0["lw"], r1["lw"])
    np.testing.assert_allclose(r0["x"].numpy(), pt.x.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r0["lw"].numpy(), lw.numpy(), rtol=1e-4, atol=1e-4)
    from oracle.numerical import effective_sample_size
    assert abs(r0["ess"] - float(effective_sample_size(lw))) < 1e-4
    assert abs(r0["log_Z"] - float(torch.logsumexp(lw.double(), 0) - np.log(total))) < 1e-4


# ---- Metropolis: the noise-scaling rule on the acceptance of ALL chains, deferred to ONE gather per call (metropolis.py:68-73) ----
class _OracleMetropolisShardBackend:
    """Stand-in for parallel.HipShardBackend with a Metropolis operator: the oracle's transitions with their own rule off
    (eval_mode), the block sums of min(1, acceptance) restated the way k_metropolis publishes them (fp32, per 16-chain block,
    row order; slab layout of fabhip_metropolis_partials_floats), the rule restated as k_metropolis_adapt_gathered."""
    tuning, hmc = True, False

    def __init__(self, n_updates=3):
        sys.path.insert(0, ROOT)
        from oracle import ais as oais, flow as oflow, targets as otgt
        self.oais, self.D, self.M, self.nu = oais, 2, 4, n_updates
        torch.manual_seed(0)
        self.nf = oflow.make_realnvp(self.D, 2, 5)
        oflow.randomize_last_layers(self.nf, 0.05, 1)
        torch.manual_seed(0)
        self.target = otgt.GMM(self.D, 8, 6.0, 0.5)
        self.op = oais.Metropolis(self.M, self.D, self.nf.log_prob, self.target.log_prob, n_updates, alpha=2.0, p_target=False,
                                  max_step_size=3.0, min_step_size=1.0, adjust_step_size=True, eval_mode=True)
        self.betas = oais.beta_schedule(self.M)
        self.decisions = []

    n_transitions = property(lambda self: self.M)

    def run_metropolis_deferred(self, b, eps0, noise_a, noise_b):
        from oracle.ais import Point
        oais, op = self.oais, self.op
        x, lq0 = (t.detach() for t in self.nf.sample_eps(eps0))
        pt = oais.create_point(x, self.nf.log_prob, self.target.log_prob, with_grad=False)
        lw = (oais.intermediate_log_prob(pt, self.betas[1], 2.0, False) - lq0).detach()
        nblk = (b + 15) // 16
        slab = torch.zeros(self.M, self.nu * nblk + 1)
        for j in range(1, self.M + 1):
            prev = oais.intermediate_log_prob(pt, self.betas[j], 2.0, False)              # never refreshed (:53)
            for n in range(self.nu):
                x_prop = pt.x + noise_a[j - 1][n, :b] * op.noise_scalings[j - 1, n]
                prop = oais.create_point(x_prop, self.nf.log_prob, self.target.log_prob, with_grad=False)
                acc = torch.exp(oais.intermediate_log_prob(prop, self.betas[j], 2.0, False) - prev)
                acc = torch.nan_to_num(acc, nan=0.0, posinf=0.0, neginf=0.0)
                accept = acc > noise_b[j - 1][n, :b]
                pt[accept] = prop[accept]
                contrib = torch.clamp_max(acc, 1).float()
                for k in range(nblk):
                    s = torch.tensor(0.0)
                    for v in contrib[16 * k:16 * k + 16]:
                        s = s + v
                    slab[j - 1, n * nblk + k] = s
            slab[j - 1, self.nu * nblk] = float(b)
            num = oais.intermediate_log_prob(pt, self.betas[j + 1], 2.0, False)
            den = oais.intermediate_log_prob(pt, self.betas[j], 2.0, False)
            lw = lw + (num - den)
        return pt, lw, slab.reshape(-1)

    def adapt_metropolis(self, gathered, world, b):
        nblk = (b + 15) // 16
        slabs = gathered.reshape(world, self.M, self.nu * nblk + 1)
        for j in range(self.M):
            for n in range(self.nu):
                s, nv = torch.tensor(0.0), 0
                for r in range(world):                     # ranks, then blocks, in order
                    for k in range(nblk):
                        s = s + slabs[r, j, n * nblk + k]
                    nv += int(slabs[r, j, self.nu * nblk])
                up = bool(s / torch.tensor(float(nv)) > 0.65)
                sc = self.op.noise_scalings
                sc[j, n] = sc[j, n] * 1.05 if up else sc[j, n] / 1.05
                self.decisions.append((up, float(s)))


def _metropolis_noise(total, D, M, nu):
    g = torch.Generator().manual_seed(13)
    eps0 = torch.randn(total, D, generator=g)
    na = torch.randn(M, nu, total, D, generator=g)
    na[:, :, total // 2:] *= 3.0                          # the second shard proposes wider and accepts less often
    nb = torch.rand(M, nu, total, generator=g)
    return eps0, na, nb


def _worker_metropolis(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    be = _OracleMetropolisShardBackend()
    sh = parallel.ShardedAnnealedImportanceSampler(backend=be)
    b = total // world
    eps0, na, nb = _metropolis_noise(total, be.D, be.M, be.nu)
    sl = slice(rank * b, (rank + 1) * b)
    x, lw, lq = sh.sample_and_log_weights(total, eps0=eps0[sl], noise_a=na[:, :, sl].contiguous(),
                                          noise_b=nb[:, :, sl].contiguous())
    torch.save({"x": x, "lw": lw, "sc": be.op.noise_scalings.clone(), "dec": be.decisions, "n_gathers": sh.n_slab_gathers},
               out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_metropolis_run_adapts_on_all_chains_with_one_gather_per_call(tmp_path):
    """VERDICT r3 missing #3: two gloo ranks x 32 chains against the oracle's OWN Metropolis (its rule on, all 64 chains in one
    process, same noise rows): the noise scalings after the call are the single-process ones bit for bit on both ranks, from
    exactly ONE slab gather - and the shards are built so that a per-rank rule would have decided differently somewhere."""
    world, total = 2, 64
    out = str(tmp_path / "m")
    mp.spawn(_worker_metropolis, args=(world, _free_port(), total, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    torch.set_num_threads(1)
    be = _OracleMetropolisShardBackend()
    oais = be.oais
    ref = oais.Metropolis(be.M, be.D, be.nf.log_prob, be.target.log_prob, be.nu, alpha=2.0, p_target=False, max_step_size=3.0,
                          min_step_size=1.0, adjust_step_size=True)           # the reference's rule, on the whole batch
    eps0, na, nb = _metropolis_noise(total, be.D, be.M, be.nu)
    x, lq0 = (t.detach() for t in be.nf.sample_eps(eps0))
    pt = oais.create_point(x, be.nf.log_prob, be.target.log_prob, with_grad=False)
    lw = (oais.intermediate_log_prob(pt, be.betas[1], 2.0, False) - lq0).detach()
    for j in range(1, be.M + 1):
        pt = ref.transition(pt, j, be.betas[j], na[j - 1], nb[j - 1])
        lw = lw + (oais.intermediate_log_prob(pt, be.betas[j + 1], 2.0, False) - oais.intermediate_log_prob(pt, be.betas[j], 2.0, False))
    for r in (r0, r1):
        assert torch.equal(r["sc"], ref.noise_scalings)
        assert r["n_gathers"] == 1
        assert len(r["dec"]) == be.M * be.nu
    assert r0["dec"] == r1["dec"]
    # what each rank alone would have decided: its own slab through the same rule
    alone = []
    for rank in range(world):
        b1 = _OracleMetropolisShardBackend()
        sl = slice(rank * 32, (rank + 1) * 32)
        _, _, slab = b1.run_metropolis_deferred(32, eps0[sl], na[:, :, sl].contiguous(), nb[:, :, sl].contiguous())
        b1.adapt_metropolis(slab, 1, 32)
        alone.append([d[0] for d in b1.decisions])
    assert alone[0] != alone[1], "test set-up: the two shards must disagree on some (transition, update)"
    assert torch.equal(r0["x"], r1["x"]) and torch.equal(r0["lw"], r1["lw"])
    np.testing.assert_allclose(r0["x"].numpy(), pt.x.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r0["lw"].numpy(), lw.numpy(), rtol=1e-4, atol=1e-4)


def test_one_op_form_is_chosen_only_when_the_group_resolves_from_cxx():
    """The tuned sharded call runs as ONE op only if c10d's registry knows the process group by name - decided once, before
    anything is enqueued, from state every rank shares; otherwise (and for backends without the op) the Python-stepped loop."""
    class _B:
        hmc, tuning = True, True
        def __init__(self, ok): self.ok, self.asked = ok, 0
        def one_op_available(self, group=None): self.asked += 1; return self.ok
        def run_tuned(self, *a): raise AssertionError("not reached in this test")
    for ok in (True, False):
        be = _B(ok)
        sh = parallel.ShardedAnnealedImportanceSampler(backend=be, one_op=True)
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert sh._use_one_op() is ok and sh._use_one_op() is ok and be.asked == 1      # asked once
            assert (len(w) == 1) == (not ok)
    assert parallel.ShardedAnnealedImportanceSampler(backend=_B(True), one_op=False)._use_one_op() is False
    assert parallel.ShardedAnnealedImportanceSampler(backend=_OracleShardBackend(), one_op=True)._use_one_op() is False   # no run_tuned


# ---- uneven splits (tuning frozen): the particle gather pads every shard to the largest one ---------------------------------
class _FrozenStubBackend:
    """Rank-local stand-in with tuning frozen: chains are rows [global index, rank], log_w = global index."""
    tuning = False

    def run_fused(self, b, eps0=None, noise_a=None, noise_b=None):
        from fab_torch_amd.point import Point
        r = dist.get_rank() if dist.is_initialized() else 0
        first = sum(parallel.shard_sizes(self.total, dist.get_world_size() if dist.is_initialized() else 1)[:r])
        idx = torch.arange(first, first + b, dtype=torch.float32)
        x = torch.stack([idx, torch.full_like(idx, float(r))], dim=1)
        return Point(x, -idx, idx, None, None), idx.clone()


def _worker_uneven(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = _FrozenStubBackend(); be.total = total
    sh = parallel.ShardedAnnealedImportanceSampler(backend=be)
    assert sh.local_batch(total) == parallel.shard_sizes(total, world)[rank]
    x, lw, lq = sh.sample_and_log_weights(total)                      # compact: exactly the chains that exist
    xf, lwf, _ = sh.sample_and_log_weights(total, compact=False)      # fixed size: world * largest shard, padding as -inf rows
    be.tuning = True
    try:
        sh.local_batch(total)
        refused = False
    except ValueError:
        refused = True
    torch.save({"x": x, "lw": lw, "xf": xf, "lwf": lwf, "refused": refused, "ess": float(sh.logging_info["ess_ais"])}, out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_split_with_tuning_frozen_pads_the_gather_and_is_refused_with_tuning_on(tmp_path):
    world, total = 2, 7                                              # shards of 4 and 3 chains
    out = str(tmp_path / "u")
    mp.spawn(_worker_uneven, args=(world, _free_port(), total, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    for r in (r0, r1):
        assert r["x"].shape == (total, 2) and torch.equal(r["x"][:, 0], torch.arange(total, dtype=torch.float32))
        assert torch.equal(r["x"][:, 1], torch.tensor([0., 0, 0, 0, 1, 1, 1])) and torch.equal(r["lw"], r["x"][:, 0])
        assert r["xf"].shape == (8, 2) and torch.isinf(r["lwf"][7]) and r["lwf"][7] < 0 and torch.equal(r["lwf"][:7], r["lw"])
        assert r["refused"]
    w = torch.exp(r0["lw"].double())
    assert abs(r0["ess"] - float(w.sum() ** 2 / (w ** 2).sum() / total)) < 1e-9        # normalised by the chains that exist
