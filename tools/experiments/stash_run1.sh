#!/bin/bash
# first GPU run of the LDS stash (FABHIP_OPT_R4_STREAM = 3): bit-identity against option 2, stage stamps, bench both ways
cd /root/repo; mkdir -p gpurun_out/stash
timeout 900 python -m pytest tests/test_gpu_hmc_shapes.py -x -q -m gpu -k "lds_prefetch or fused_stage_four or one_launch_tail" > gpurun_out/stash/pytest1.txt 2>&1; tail -5 gpurun_out/stash/pytest1.txt
timeout 300 python tools/timeline_r4.py > gpurun_out/stash/tl_stash.txt 2>&1; cat gpurun_out/stash/tl_stash.txt | tail -16
FABHIP_R4_STREAM=2 timeout 300 python tools/timeline_r4.py > gpurun_out/stash/tl_ring.txt 2>&1; cat gpurun_out/stash/tl_ring.txt | tail -16
timeout 600 python bench.py > gpurun_out/stash/bench.json 2> gpurun_out/stash/bench.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/stash/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['ms_per_launch'])
PY
FABHIP_R4_STREAM=2 timeout 600 python bench.py > gpurun_out/stash/bench_ring.json 2>> gpurun_out/stash/bench.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/stash/bench_ring.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['ms_per_launch'])
PY
