set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_hmc_shapes.py -x -q -m gpu -k "small_tiles or eight_chain" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "headline_architecture" 2>&1 | tail -4
timeout 400 python tools/time_hmc_shapes.py 2>&1 | grep "W="
timeout 300 python tools/timeline_r8.py 2048 2>&1 | tail -13
