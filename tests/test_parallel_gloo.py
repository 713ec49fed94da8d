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
