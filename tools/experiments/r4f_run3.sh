cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_hmc_shapes.py -x -q 2>&1 | tail -12 | tee gpurun_out/r4f/pytest_fast.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4f/bench_fast.json 2> gpurun_out/r4f/bench_fast.err; tail -3 gpurun_out/r4f/bench_fast.err
python - <<'PY'
import json
for ln in open('gpurun_out/r4f/bench_fast.json'):
    if ln.startswith('{'):
        d=json.loads(ln); print('value',d['value'],'fast',d['fast_mode']['value'],d['fast_mode']['speedup_vs_value'],d['fast_mode']['ess_trained'])
PY
