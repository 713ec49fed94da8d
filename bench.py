#!/usr/bin/env python
"""bench.py — AIS samples/s on the BASELINE headline workload.

Workload (BASELINE.json metric / north_star): ManyWell-32 target, RealNVP flow of the
reference's ManyWell-32 architecture (10 x [MLP 16-320-320-32 coupling + InvertibleAffine]), 1024 chains per
GPU, 8 intermediate distributions (linear beta), HMC with 5 leapfrog steps / 1 outer step, alpha = 2,
p_target = False, step-size tuning ON (as in training).  Random-init weights (seeded; last coupling
layer re-drawn N(0, 0.01^2) so log-dets are non-trivial), synthetic noise drawn on the device inside the
timed region.  A "step" is one `AnnealedImportanceSampler.sample_and_log_weights(1024)` call per GPU
(+ the RCCL all-gather of the particles when --gpus > 1).

Prints ONE JSON line (rank 0): value = chains produced by all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, K_LAYERS, NODES, B_PER_GPU, M, L = 32, 10, 10, 1024, 8, 5
ALPHA = 2.0
EPS_INIT = 0.2                       # close to the tuned value (0.18 for the p^2/q target)
F_FWD = K_LAYERS * 2 * (16 * 320 + 320 * 320 + 2 * 320 * 16) + 2 * K_LAYERS * D * D    # 2 375 680 flop / sample / pass
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def build_flow_state(seed=0):
    """Seeded RealNVP parameters (CPU tensors, identical on every rank and for the CPU baseline)."""
    from fab_torch_amd.flow import RealNVP
    torch.manual_seed(seed)
    flow = RealNVP(D, K_LAYERS, NODES)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for l1, l2, l3, aff in flow._layers():
            l3.weight.copy_(torch.randn(l3.weight.shape, generator=g) * 0.01)
            l3.bias.copy_(torch.randn(l3.bias.shape, generator=g) * 0.01)
    return flow


CPU_THREADS = 16                     # the thread count the oracle's eager CPU path is timed at (best of the 8/16/32/64 sweeps of r1-r3)


def cpu_baseline(flow_state, n_calls=20, n_warm=5):
    """The oracle (PyTorch-CPU restatement of the reference path) timed on this box's host cores on the same workload at a
    FIXED thread count (CPU_THREADS): median of n_calls = 20 full AIS calls of 1024 chains after n_warm = 5 warm-up calls
    (SURVEY 8d's protocol; ~20 s of CPU work).  A one-call-each sweep over other thread counts is reported next to it as
    information, never as `value`."""
    from oracle import ais as oais, flow as oflow, targets as otgt
    nf = oflow.make_realnvp(D, K_LAYERS, NODES)
    nf.load_state_dict(flow_state)
    target = otgt.ManyWell(D)
    hmc = oais.HMC(M, D, nf.log_prob, target.log_prob, alpha=ALPHA, p_target=False, epsilon=EPS_INIT, L=L)
    ais = oais.AIS(lambda e: tuple(t.detach() for t in nf.sample_eps(e)), nf.log_prob, target.log_prob, hmc,
                   False, ALPHA, M)

    def call():
        eps0 = torch.randn(B_PER_GPU, D)
        noise_p = torch.randn(M, 1, B_PER_GPU, D)
        noise_e = torch.empty(M, 1, B_PER_GPU).exponential_()
        t0 = time.perf_counter()
        ais.sample_and_log_weights(eps0, noise_p, noise_e)
        return time.perf_counter() - t0

    threads = min(CPU_THREADS, os.cpu_count() or CPU_THREADS)
    torch.set_num_threads(threads)
    for _ in range(n_warm):
        call()
    dts = sorted(call() for _ in range(n_calls))
    dt = dts[n_calls // 2]
    sweep = {}
    for nt in (8, 32, 64):
        if nt <= (os.cpu_count() or 8) and nt != threads:
            torch.set_num_threads(nt)
            call()
            sweep[str(nt)] = B_PER_GPU / call()
    torch.set_num_threads(threads)
    return {"value": B_PER_GPU / dt, "unit": "AIS samples/s", "cores": threads, "kind": "port",
            "sample": f"median of {n_calls} calls of sample_and_log_weights({B_PER_GPU}) after {n_warm} warm-up calls, {threads} threads "
                      f"(fixed), fp32, oracle/ = PyTorch-CPU eager + autograd per leapfrog",
            "sec_per_call": dt, "sec_per_call_min_max": [dts[0], dts[-1]], "host_cpus": os.cpu_count(),
            "thread_sweep_samples_per_s": sweep}


PEAK_HBM_TBPS = 8.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def _lib_srchash():
    """Content hash of the sources libfabhip.so was built from (fab_torch_amd/_build.py stamp): identifies the kernels."""
    try:
        with open(os.path.join(ROOT, "fab_torch_amd", "libfabhip.so.srchash")) as f:
            return f.read().split()[0]
    except (OSError, IndexError):
        return "unknown"


def _event_time(fn, n=20, warm=3):
    """median seconds per call, HIP events on torch's current stream (the stream the ops enqueue on)."""
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ms[n // 2] * 1e-3


def resample_rooflines(dev, log2n=26):
    """HBM rooflines of the resample path at N = 2^26 log-weights ~ N(0, 3^2) (seed 0), HIP-event timed in-process:
    (1) k_scan_fixed_lds alone - the decoupled-look-back fixed-point CDF scan: 4N read + 8N written = 12N algorithmic
    bytes; (2) the whole fabhip_resample_systematic call (max pass, tile sums, prefix, fused emit): algorithmic bytes
    4N (log_w in) + 8N (idx out) = 12N, actual traffic 20N (log_w is read three times)."""
    from fab_torch_amd import _ops
    ops = _ops.load()
    N = 1 << log2n
    g = torch.Generator(device=dev).manual_seed(0)
    lw = torch.randn(N, device=dev, generator=g) * 3
    ws = ops.fixed_cdf(lw, None)                                   # max + scan once: leaves the max in the workspace
    t_scan = _event_time(lambda: ops.fixed_cdf(lw, ws))            # memset of the descriptors (~5 us) + the scan kernel
    t_sys = _event_time(lambda: ops.resample_systematic(lw, 0.3, N))
    u = torch.rand(N, device=dev, generator=g, dtype=torch.float64)
    t_mult = _event_time(lambda: ops.resample_multinomial(lw, u), n=8, warm=2)
    del u
    rows = []
    for name, t, alg, traffic in (("k_scan_fixed_lds (fixed-point CDF scan, decoupled look-back)", t_scan, 12 * N, 12 * N),
                                  ("fabhip_resample_systematic end to end (max, tile sums, prefix, fused emit)", t_sys,
                                   12 * N, 20 * N),
                                  ("fabhip_resample_multinomial end to end (scan + per-draw 16-ary search of the CDF: random 128-byte "
                                   "lines, not a stream)", t_mult, 20 * N, None)):
        ach = alg / t / 1e12
        rows.append({"bound": "hbm", "kernel": name, "N": N, "achieved": ach, "peak": PEAK_HBM_TBPS, "unit": "TB/s",
                     "frac": ach / PEAK_HBM_TBPS, "traffic": traffic, "traffic_TBps": (traffic / t / 1e12) if traffic else None,
                     "us_per_call": t * 1e6, "algorithmic_bytes": alg})
    del lw, ws
    torch.cuda.empty_cache()
    return rows


def trained_flow_ess(dev, fast=False):
    """"Meaningful ESS" row: the committed small TRAINED flow (tests/golden/g13: ManyWell-6, trained with
    the reference's PrioritisedBufferTrainer) and the reference's own evaluation AIS call on it (1024 chains, target p,
    frozen step sizes, captured noise): the HIP path on the identical noise must reproduce its ESS (north_star: within 1 %)."""
    import numpy as np
    import fab_torch_amd as fa
    path = os.path.join(ROOT, "tests", "golden", "g13_trained_flow_mw6.npz")
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    D, K, nodes, M, L = int(g["D"]), int(g["K"]), int(g["nodes"]), int(g["M"]), int(g["L"])
    flow = fa.RealNVP(D, K, nodes)
    flow._nf_model.load_state_dict({k[len("flow."):]: torch.tensor(v) for k, v in g.items() if k.startswith("flow.")})
    flow = flow.to(dev).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    out = {"fixture": "tests/golden/g13_trained_flow_mw6.npz", "chains": int(g["B"])}
    for tag, p_target in (("p", True), ("g", False)):
        hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=float(g["alpha"]), p_target=p_target,
                                       epsilon=1.0, L=L, eval_mode=True).to(dev)
        with torch.no_grad():
            hmc.epsilons.copy_(torch.tensor(g["epsilons"])); hmc.common_epsilon.copy_(torch.tensor(g["common_epsilon"]))
        ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target, float(g["alpha"]), M)
        T = lambda k: torch.tensor(g[k]).to(dev)      # noqa: E731
        with fa.fast_mode(fast):
            ais.sample_and_log_weights(int(g["B"]), eps0=T(f"{tag}_eps0"), noise_a=T(f"{tag}_noise_p"),
                                       noise_b=T(f"{tag}_noise_e"))
        info = ais.get_logging_info()
        ref = float(g[f"{tag}_ess_ais"])
        out["target_" + tag] = {"ess_ais_hip": info["ess_ais"], "ess_ais_reference": ref,
                                "rel_diff": abs(info["ess_ais"] - ref) / ref, "ess_flow_hip": info["ess_base"],
                                "ess_flow_reference": float(g[f"{tag}_ess_base"]), "log_Z_hip": info["log_Z"],
                                "log_Z_reference": float(g[f"{tag}_log_Z"])}
    out["log_Z_exact"] = float(target.log_Z)
    return out


WORKLOADS = {
    # name: (chains per GPU, coupling layers, intermediate distributions)
    "headline": (1024, 10, 8),      # BASELINE.json metric / north_star: ManyWell-32, 1024 chains, M = 8
    "cfg4": (2048, 12, 12),         # BASELINE cfg 4's per-GPU shape: 16384 chains over 8 GPUs, 12 layers, M = 12
}


def spline_cfg3(dev):
    """BASELINE cfg 3 with the flow family it names (reported next to the headline, not `value`): ManyWell-32, spline flow
    12 x (hidden 256, 8 bins), 2048 chains, 12 intermediate distributions, HMC L = 5, through the fused spline AIS call
    (fabhip_spline_ais_run; the density + gradient kernel is k_spline_logprob_r8: 8 chains per workgroup, spline_r8.h)."""
    import fab_torch_amd as fa
    D, L, H, M, B, LF = 32, 12, 256, 12, 2048, 5
    torch.manual_seed(0)
    flow = fa.make_wrapped_normflow_spline(D, L, H, (), 5.0).to(dev).requires_grad_(False)
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] % 25 == 0:
                p.add_(0.02 * torch.randn_like(p))
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=0.1, L=LF).to(dev)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M)
    x, _ = flow.sample_and_log_prob((B,))
    t_eval = _event_time(lambda: flow.log_prob_and_grad(x), n=30, warm=5)
    for _ in range(2):
        ais.sample_and_log_weights(B)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ais.sample_and_log_weights(B)
    torch.cuda.synchronize(dev)
    t_call = (time.perf_counter() - t0) / n
    flop = 4.0 * (64 * H + 2 * H * H + 512 * H) * L * B            # conditioner GEMMs as executed, forward + reverse
    flop_alg = 4.0 * (16 * H + 2 * H * H + 400 * H) * L * B        # algorithmic: 16 identity features in, 25 x 16 parameters out
    stream = 2 * 4 * 272 * 1024 * L                                # weight tiles one workgroup streams per evaluation
    return {"workload": "cfg3: ManyWell-32, spline flow 12 x (hidden 256, 8 bins), 2048 chains, M = 12, HMC L = 5",
            "value": B / t_call, "unit": "AIS samples/s", "ms_per_call": t_call * 1e3,
            "density_grad_evals_per_call": M * LF + 1, "ms_per_density_grad": t_eval * 1e3,
            "kernel": "k_spline_logprob_r8<2, 2, true, true> (8 chains per workgroup, stream without zero tiles, v_mfma_f32_4x4x1, 256 workgroups)",
            "achieved_TFLOPs": flop_alg / t_eval / 1e12, "frac_fp32_mfma_peak": flop_alg / t_eval / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "flop_per_eval_algorithmic": flop_alg, "flop_per_eval_as_executed": flop,
            "as_executed": {"achieved_TFLOPs": flop / t_eval / 1e12, "frac_fp32_mfma_peak": flop / t_eval / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                            "note": "64-wide input and 512-wide output tiles for 16 / 400 real columns: 17 % padding"},
            "weight_stream_bytes_per_workgroup": stream,
            "ess_ais": float(ais.get_logging_info()["ess_ais"])}


def trainer_iteration(dev):
    """The training half next to the sampler: one iteration of fab/train_with_prioritised_buffer.py:138-216 on
    the reference's ManyWell-32 recipe (experiments/config/many_well.yaml: batch 2048, M = 4, HMC L = 5, 8 minibatches of 2048 from
    a 512 000-entry prioritised buffer, alpha = 2) - tools/bench_trainer.py: N back-to-back trainer steps between two device
    synchronisations, the AIS call of the same sampler alone by HIP events, the two training kernels stand-alone."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_trainer
    out, trainer = bench_trainer.measure(dev)
    out["kernels"] = bench_trainer.kernel_rows(trainer, dev)
    out["profile"] = "profiles/r6/trainer_kernel_stats_rocprofv3.csv + trainer_iteration_timeline.txt (tools/trace_trainer.sh: the same script under rocprofv3 --kernel-trace --stats)"
    del trainer
    torch.cuda.empty_cache()
    return out


def _relaunch_under_torchrun(args):
    """`python bench.py --gpus N` started directly (no torchrun environment): start N ranks of this same script, one per
    GPU, under torch.distributed.run on 127.0.0.1 and hand its exit code back.  (The driver's own torchrun command line
    arrives here with WORLD_SIZE set and skips this.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


class _StubSampler:
    """Launcher self-test (`--stub-step`, tests/test_bench_launcher.py): a CPU stand-in for the rank-local AIS call, so
    that the re-exec under torch.distributed.run, the rendezvous, the barrier / max-over-ranks timing, the particle
    all-gather and the JSON line can run on a box without GPUs.  Never a measurement: the line says so in `data`."""

    def __init__(self, dev):
        self.dev = dev

    def sample_and_log_weights(self, total, compact=False):
        from fab_torch_amd import parallel
        b = total // parallel._world()
        g = torch.Generator().manual_seed(parallel._rank())
        x = torch.randn(b, D, generator=g).to(self.dev)
        lw = torch.randn(b, generator=g).to(self.dev)
        return parallel.gather_particles(x, lw, lw.clone(), b, compact=compact)


def main():
    global B_PER_GPU, K_LAYERS, M, F_FWD
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline")
    ap.add_argument("--chains-per-gpu", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--M", type=int, default=None)
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch_under_torchrun(args)

    B_PER_GPU, K_LAYERS, M = WORKLOADS[args.workload]
    B_PER_GPU = args.chains_per_gpu or B_PER_GPU
    K_LAYERS = args.layers or K_LAYERS
    M = args.M or M
    F_FWD = K_LAYERS * 2 * (16 * 320 + 320 * 320 + 2 * 320 * 16) + 2 * K_LAYERS * D * D
    custom = (B_PER_GPU, K_LAYERS, M) != WORKLOADS[args.workload]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # FABHIP_BENCH_BACKEND=gloo: ranks may share a GPU (2-process check on a 1-GPU box; payloads staged through the host)
    backend = os.environ.get("FABHIP_BENCH_BACKEND", "nccl")
    if args.stub_step:
        backend, dev = "gloo", torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the fab_torch_amd hot path has no CPU fallback")
        n_dev = torch.cuda.device_count()
        if distributed and backend == "nccl" and world > n_dev:
            raise SystemExit(f"bench.py --gpus {world}: this node has {n_dev} GPU(s) (RCCL needs one GPU per rank)")
        dev = torch.device("cuda", local_rank % n_dev)
        torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that never arrives (dead GPU, wrong device mapping, RCCL bootstrap over xGMI failing) must end this run with an
        # error line, not hold the node until the lease expires: bounded rendezvous + collective timeouts, asynchronous error
        # handling that tears the process down (tools/scale_run.sh reads the exit codes)
        import datetime
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        tmo = datetime.timedelta(seconds=int(os.environ.get("FABHIP_BENCH_PG_TIMEOUT", "180")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
        except Exception as e:                             # noqa: BLE001 - reported, then fatal
            print(json.dumps({"error": f"init_process_group({backend}) failed on rank {rank}/{world}: {e!r}"}), flush=True)
            raise SystemExit(3)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        sync()

    def max_over_ranks(t):
        if not distributed:
            return t
        tt = torch.tensor([t], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    if args.stub_step:
        sh = _StubSampler(dev)
        step = lambda: sh.sample_and_log_weights(world * B_PER_GPU, compact=False)      # noqa: E731
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"metric": "AIS samples/sec (+ESS), ManyWell-32, 8 intermediate dists",
                              "value": world * B_PER_GPU * args.steps / elapsed, "unit": "AIS samples/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                              "data": "stub (launcher self-test on CPU ranks, not a measurement)",
                              "config": {"workload": "launcher self-test", "chains_per_gpu": B_PER_GPU,
                                         "global_chains": world * B_PER_GPU},
                              "ranks": world, "backend": backend, "gathered_rows": int(out[0].shape[0]),
                              "process_group_ranks": (dist.get_world_size() if distributed else 1),
                              "rccl_ranks": 0, "collectives_per_step": 1 if distributed else 0}))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    import fab_torch_amd as fa
    from fab_torch_amd import parallel

    flow = build_flow_state(0)
    flow_state = {k: v.clone() for k, v in flow._nf_model.state_dict().items()}
    flow = flow.to(dev).requires_grad_(False)
    target = fa.ManyWellEnergy(D)
    hmc = fa.HamiltonianMonteCarlo(M, D, flow.log_prob, target.log_prob, alpha=ALPHA, p_target=False,
                                   epsilon=EPS_INIT, n_outer=1, L=L).to(dev)
    ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=ALPHA,
                                       n_intermediate_distributions=M)
    # N > 1: chains sharded over the ranks with the SINGLE-DEVICE step-size rule (one acceptance-slab all-gather per
    # transition while tuning is on) + one all-gather of the particles; fixed-size result, no host synchronisation
    sharded = parallel.ShardedAnnealedImportanceSampler(ais) if distributed else None
    torch.manual_seed(1234 + rank)          # per-rank noise streams

    def step():
        if distributed:
            return sharded.sample_and_log_weights(world * B_PER_GPU, compact=False)   # (dropped chains: log_w = -inf rows)
        pt, log_w = ais.sample_and_log_weights(B_PER_GPU)
        return pt.x, log_w, pt.log_q

    for _ in range(200):                        # untimed, same count on every rank (step() holds collectives): a fresh
        step()                                  # box needs ~1 s of work to reach steady clocks (measured in round 3: the first
    sync()                                      # process on a box read 4 % low after 30 such steps)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    info = ais.get_logging_info()                   # (N > 1: this rank's chains; `ess_gathered` below is the whole set)
    if distributed:
        info["log_Z"] = float(sharded.logging_info["log_Z"])
    ess_all = float(fa.effective_sample_size(out[1]).item())
    slab_gathers = sharded.n_slab_gathers if distributed else 0
    # second row of SURVEY.md section 8d: the same K steps with step-size tuning frozen (evaluation mode)
    hmc.set_eval_mode(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_eval = max_over_ranks(time.perf_counter() - t0)
    slab_gathers_eval = sharded.n_slab_gathers if distributed else 0
    hmc.set_eval_mode(False)
    # the particle all-gather alone (the one data-path collective), HIP events on the collective's stream
    gather_us = None
    if distributed:
        px, plw, plq = out[0][:B_PER_GPU].contiguous(), out[1][:B_PER_GPU].contiguous(), out[2][:B_PER_GPU].contiguous()
        gather_us = _event_time(lambda: parallel.gather_particles(px, plw, plq, B_PER_GPU, compact=False)) * 1e6
    # third row: FAST MODE (bf16 W x W GEMMs in the transition kernels; NOT the parity path, never `value`) on the same
    # workload, step-size tuning on ("an fp32 parity mode and a fast mode, report both")
    saved = {k: v.clone() for k, v in hmc.state_dict().items()}
    with fa.fast_mode():
        for _ in range(10):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed_fast = max_over_ranks(time.perf_counter() - t0)
        info_fast = ais.get_logging_info()
    hmc.load_state_dict(saved)

    # ---- roofline of the dominant kernel (k_hmc_step): live HIP-event timing on the launch stream -----
    roof = None
    if rank == 0:
        # The kernel the timed step runs: M back-to-back launches of the transition kernel with the step-size rule in its last
        # wave, enqueued by ONE op (torch.ops.fabhip.ais_phase = fabhip_ais_phase, the call `step` itself makes) and bracketed
        # by HIP events on torch's current stream (= the stream handed to the C ABI).  Per launch = (chain initialisation +
        # transitions 1 .. M) - (chain initialisation alone), medians of 10 calls each, / M  (rounds 1 - 4
        # timed a stand-alone hmc.transition, which adds a k_hmc_adapt launch and a dispatch per transition).
        saved_roof = {k: v.clone() for k, v in hmc.state_dict().items()}
        shard = parallel.HipShardBackend(ais)

        def time_transition(n_chains, n_t=10):
            st = shard._state(n_chains)
            t_init = _event_time(lambda: shard._phase(st, 1, 1, 0), n=n_t, warm=3)
            t_full = _event_time(lambda: shard._phase(st, 1, 1, M, tune=True), n=n_t, warm=3)
            return (t_full - t_init) / M

        t_kernel = time_transition(B_PER_GPU)
        flop = B_PER_GPU * L * 2 * F_FWD                  # flow fwd + d/dx per leapfrog (target flops ignored)
        ach = flop / t_kernel / 1e12
        from fab_torch_amd import _ops
        shape = int(_ops.load().get_option(_ops.OPT_TILE_SHAPE))
        r4 = shape == 4 or (shape == 0 and B_PER_GPU <= 1152)               # 4-chain tiles (flow_r4.h) below 1153 chains
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        r8 = shape == 8 or (shape == 0 and 1152 < B_PER_GPU <= 8 * n_cu)    # 8-chain tiles (flow_r8.h) up to 8 chains per CU
        n_wg = (B_PER_GPU + 3) // 4 if r4 else ((B_PER_GPU + 7) // 8 if r8 else (B_PER_GPU + 15) // 16)
        kname = "k_hmc_step_r4<5> (4 chains per workgroup, v_mfma_f32_4x4x1; step-size rule in its last wave)" if r4 else \
            ("k_hmc_step_r8<5, true> (8 chains per workgroup, fused stages, v_mfma_f32_4x4x1; step-size rule in its last wave)" if r8 else
             "k_hmc_step<5> (+ k_hmc_adapt, ~2 us)")
        roof = {"bound": "mfma", "kernel": kname, "achieved": ach,
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                "traffic": None, "ms_per_launch": t_kernel * 1e3, "flop_per_launch": flop,
                "timing": "HIP events around fabhip_ais_phase(init + M transitions) minus (init), / M: the launches of the timed step",
                "workgroups": n_wg, "frac_of_occupied_cus": ach / (PEAK_FP32_MFMA_TFLOPS * min(n_wg, 256) / 256)}
        # HBM traffic per launch: separate rocprofv3 --pmc passes of tools/prof_hmc.py (same kernel, same shape),
        # summarised by tools/pmc_summary.py and committed; not collectable from inside this process.
        if r4:
            # the binding resource of the 4-chain kernel is not the matrix pipe but each CU's L2 -> VGPR weight stream: every
            # workgroup reads the whole r4 image once per flow evaluation (forward + reverse sweep)
            wp = 64 * ((D * NODES + 63) // 64)
            per_pair = 4 * (2 * 32 * 64 + 16 * wp + 2 * wp * wp + wp * 32 + wp * 16 + 32 * wp)     # bytes per layer pair (D = 32)
            stream = per_pair * K_LAYERS * L
            clk = 2.09e9                                     # shader clock under this kernel (s_memtime stamps against HIP events)
            roof["weight_stream"] = {"bytes_per_workgroup_per_launch": stream, "GBps_per_cu": stream / t_kernel / 1e9,
                                     "B_per_clk_per_cu": stream / t_kernel / clk, "path_B_per_clk_per_cu": 58.0,
                                     "frac_of_path": stream / t_kernel / clk / 58.0,
                                     "note": "tools/ubench/stream2.hip (profiles/r5/ubench_stream2_prefetch.txt): a stage-synchronised "
                                             "4-wave stream of this shape pulls 58 B/clk per CU from a resident L2 and 46.5 when "
                                             "every line is its XCD's first touch, as in this kernel (47.9 MB through a 4 MB L2 "
                                             "per evaluation); the kernel prefetches its stream into the L2 (DESIGN.md section 9 "
                                             "item 1); over the whole launch incl. the short stages"}
        # HBM-side traffic per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes restricted to this kernel
        # (tools/pmc_traffic.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), committed summary
        for path in (tuple(os.path.join("profiles", rnd, "hmc_step_r4_traffic_pmc_summary.json") for rnd in ("r6", "r5")) if r4 else
                     tuple(os.path.join("profiles", rnd, "hmc_step_pmc_summary.json") for rnd in ("r5", "r4"))):
            if os.path.exists(os.path.join(ROOT, path)) and args.workload == "headline" and not custom:
                with open(os.path.join(ROOT, path)) as f:
                    summ = json.load(f)
                lib_hash = _lib_srchash()
                if summ.get("lib_srchash") == lib_hash:
                    roof["traffic"] = summ.get("_derived", {}).get("hbm_bytes_per_launch")
                    roof["traffic_source"] = path + " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes; taken on this library)"
                else:                                       # counters of ANOTHER build of the kernels: not this run's traffic
                    roof["traffic"] = None
                    roof["traffic_source"] = (f"{path} was taken on library {str(summ.get('lib_srchash'))[:12]}, this run uses "
                                              f"{lib_hash[:12]}: stale, not reported (tools/pmc_traffic.sh refreshes it)")
                break
        # 16-chain tiles (k_hmc_step<5>) with one workgroup per CU (4096 chains):
        t_full = time_transition(4096)
        ach_full = 4096 * L * 2 * F_FWD / t_full / 1e12
        roof["full_chip"] = {"chains": 4096, "ms_per_launch": t_full * 1e3, "achieved": ach_full,
                             "frac": ach_full / PEAK_FP32_MFMA_TFLOPS}
        # 8-chain tiles (k_hmc_step_r8<5>) with one workgroup per CU (2048 chains: the shape of BASELINE cfg 4 per GPU):
        t_2k = time_transition(2048)
        ach_2k = 2048 * L * 2 * F_FWD / t_2k / 1e12
        roof["chains_2048"] = {"chains": 2048, "kernel": "k_hmc_step_r8<5, true> (8 chains per workgroup, fused stages)", "ms_per_launch": t_2k * 1e3,
                               "achieved": ach_2k, "frac": ach_2k / PEAK_FP32_MFMA_TFLOPS}
        hmc.load_state_dict(saved_roof)

    # ---- second roofline: the resample scan + the whole systematic resampler at N = 2^26 (HBM-bound) ----
    roof_extra, ess_trained, ess_trained_fast = None, None, None
    if rank == 0:
        roof_extra = resample_rooflines(dev)
        ess_trained = trained_flow_ess(dev)
        ess_trained_fast = trained_flow_ess(dev, fast=True)
    spline3 = spline_cfg3(dev) if (rank == 0 and world == 1 and not custom and args.workload == "headline") else None
    trainer_row = trainer_iteration(dev) if (rank == 0 and world == 1 and not custom and args.workload == "headline") else None

    bad = []
    if rank == 0:
        total = world * B_PER_GPU * args.steps
        line = {
            "metric": "AIS samples/sec (+ESS), ManyWell-32, %d intermediate dists" % M,
            "value": total / elapsed, "unit": "AIS samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s%s: ManyWell-32 AIS, RealNVP %dx(16-320-320-32)+InvAffine (seeded default init, last coupling "
                                   "Linears N(0, 0.01^2)), HMC L=5 n_outer=1, M=%d linear beta, alpha=2 (target p^2/q), "
                                   "step-size tuning on" % (args.workload, " (modified)" if custom else "", K_LAYERS, M),
                       "chains_per_gpu": B_PER_GPU, "global_chains": world * B_PER_GPU,
                       "parallelism": (f"chains sharded x{world}: single-device step-size rule ({slab_gathers} acceptance-slab "
                                       "all-gathers per step) + one particle all-gather" if world > 1 else "single GPU")},
            "ranks": world, "backend": ("rccl" if backend == "nccl" else backend) if distributed else None,
            "rccl_ranks": (dist.get_world_size() if (distributed and backend == "nccl") else 0),
            "process_group_ranks": (dist.get_world_size() if distributed else 1),
            "collectives_per_step": (slab_gathers + 1) if distributed else 0,
            "collectives_per_step_eval_mode": (slab_gathers_eval + 1) if distributed else 0,
            "lib_srchash": _lib_srchash(),
            "gathered_rows": int(out[0].shape[0]), "particle_all_gather_us": gather_us,
            "slab_all_gathers_per_step": slab_gathers,
            "value_eval_mode": total / elapsed_eval,
            "ess_ais": info["ess_ais"], "ess_gathered": ess_all, "log_Z": info["log_Z"],
            "p_accept_first": info.get("dist0_p_accept_0"),
            "roofline": roof,
            "roofline_resample": roof_extra,
            "ess_trained": ess_trained,
            "spline_cfg3": spline3,
            "trainer_iteration": trainer_row,
            "fast_mode": {
                "what": "NOT the parity path: the two 320x320 GEMMs of every coupling layer with bf16 operands (weights rounded at pack "
                        "time, activations as they are fetched; fp32 accumulation) inside the chain-initialisation and transition "
                        "kernels - up to 1152 chains on the 4-chain tiles with fused stages (v_mfma_f32_4x4x4_16b_bf16, half the "
                        "weight stream), above on the 16-chain tiles (v_mfma_f32_16x16x32_bf16); log q deviates 1e-3 .. 1e-2 from "
                        "the fp32 kernels",
                "value": total / elapsed_fast, "unit": "AIS samples/s", "ms_per_step": elapsed_fast / args.steps * 1e3,
                "speedup_vs_value": elapsed / elapsed_fast, "ess_ais": info_fast["ess_ais"], "log_Z": info_fast["log_Z"],
                "ess_trained": None if ess_trained_fast is None else
                {t: {k: ess_trained_fast[t][k] for k in ("ess_ais_hip", "ess_ais_reference", "rel_diff", "log_Z_hip")}
                 for t in ("target_p", "target_g")},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(flow_state)
            line["speedup_vs_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        # N > 1: what the line claims about the collectives is checked, not assumed (a run that silently fell back to another
        # backend, another loop form or rank-local step sizes fails here instead of producing a plausible number)
        if distributed:
            if backend == "nccl" and line["rccl_ranks"] != world:
                bad.append(f"rccl_ranks {line['rccl_ranks']} != world {world}")
            if line["process_group_ranks"] != world:
                bad.append(f"process group holds {line['process_group_ranks']} ranks, launched {world}")
            if line["collectives_per_step"] != M + 1:
                bad.append(f"tuned step issued {line['collectives_per_step']} collectives, expected M + 1 = {M + 1}")
            if line["collectives_per_step_eval_mode"] != 1:
                bad.append(f"eval-mode step issued {line['collectives_per_step_eval_mode']} collectives, expected 1")
            if line["gathered_rows"] != world * B_PER_GPU:
                bad.append(f"gathered {line['gathered_rows']} rows, expected {world * B_PER_GPU}")
        line["multi_gpu_checks"] = "ok" if not bad else bad
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()                      # rank 0 measured the roofline leg alone; leave together
        dist.destroy_process_group()
    if rank == 0 and bad:
        raise SystemExit("bench.py: multi-GPU consistency checks failed: " + "; ".join(bad))


if __name__ == "__main__":
    main()
