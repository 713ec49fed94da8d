"""`python bench.py --gpus N` must start N ranks by itself (the flag was parsed and ignored).  Runs the
launcher exactly as the driver's fallback would - no torchrun environment - with `--stub-step`: bench.py's CPU stand-in
for the rank-local AIS call (gloo), so the re-exec under torch.distributed.run, the rendezvous on 127.0.0.1, the barrier /
max-over-ranks timing, the particle all-gather and the ONE JSON line are exercised on a box without GPUs."""
import json
import os
import subprocess
import sys

from helpers import ROOT


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--stub-step",
                        "--chains-per-gpu", "64"] + extra, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_2_relaunches_itself_as_two_ranks():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"
    assert line["gathered_rows"] == 2 * 64 and line["config"]["global_chains"] == 128
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    assert "not a measurement" in line["data"]


def test_default_is_one_rank_without_a_process_group():
    line = _run([])
    assert line["n_gpus"] == 1 and line["gathered_rows"] == 64


def test_gpus_2_cfg4_line_counts_ranks_and_collectives():
    """the line of a 2-rank run states how many ranks the process group had and how many collectives a
    step issued (the stub's step = the particle all-gather alone), for BASELINE cfg 4's per-GPU workload too."""
    line = _run(["--gpus", "2", "--workload", "cfg4"])
    assert line["n_gpus"] == 2 and line["process_group_ranks"] == 2 and line["collectives_per_step"] == 1
    assert line["rccl_ranks"] == 0 and line["backend"] == "gloo"          # gloo on CPU ranks: RCCL itself needs one GPU per rank
