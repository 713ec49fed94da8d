#!/bin/bash
# first GPU run of the fused 8-chain stages: tests that touch the 8-chain kernels, then the bench line (chains_2048)
cd /root/repo; mkdir -p gpurun_out/r8f
timeout 900 python -m pytest tests/test_gpu_hmc_shapes.py -x -q -m gpu -k "eight or small_tiles or one_launch_tail" > gpurun_out/r8f/pytest1.txt 2>&1; tail -5 gpurun_out/r8f/pytest1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "headline_architecture or g14 or g15" > gpurun_out/r8f/pytest2.txt 2>&1; tail -5 gpurun_out/r8f/pytest2.txt
timeout 600 python bench.py > gpurun_out/r8f/bench.json 2> gpurun_out/r8f/bench.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r8f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['chains_2048'], d['roofline']['full_chip'])
PY
FABHIP_R4_STREAM=1 timeout 600 python bench.py > gpurun_out/r8f/bench_unfused.json 2>> gpurun_out/r8f/bench.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r8f/bench_unfused.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['chains_2048'])
PY
