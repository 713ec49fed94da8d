set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_spline.py -x -q -m gpu 2>&1 | tail -4
for b in 512 1024; do for s in 4 8; do B=$b N=3 FABHIP_TILE=$s timeout 300 python tools/bench_spline.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('B=$b tile=$s', d['log_prob_and_grad_ms'], d['ais_samples_per_s'])"; done; done
