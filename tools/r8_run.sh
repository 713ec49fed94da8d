set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_spline.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python tools/timeline_spline.py 2>&1 | tail -16
FABHIP_SPLINE_MFMA=16 timeout 300 python tools/timeline_spline.py 2>&1 | tail -3
timeout 600 python tools/bench_spline.py 2>&1 | tail -2
FABHIP_SPLINE_MFMA=16 timeout 600 python tools/bench_spline.py 2>&1 | tail -2
FABHIP_TILE=16 timeout 300 python tools/timeline_spline.py 2>&1 | tail -14
CFG=5 N=3 timeout 600 python tools/bench_spline.py 2>&1 | tail -1
CFG=5 N=3 FABHIP_SPLINE_MFMA=16 timeout 600 python tools/bench_spline.py 2>&1 | tail -1
CFG=5 N=3 FABHIP_TILE=8 timeout 600 python tools/bench_spline.py 2>&1 | tail -1
