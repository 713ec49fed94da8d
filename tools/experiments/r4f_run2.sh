cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r4f/pytest_full.txt
timeout 300 python tools/time_hmc.py 1024 2>&1 | tail -1 | tee gpurun_out/r4f/time.txt
FABHIP_R4_STREAM=1 timeout 300 python tools/time_hmc.py 1024 2>&1 | tail -1 | tee -a gpurun_out/r4f/time.txt
timeout 300 python tools/timeline_r4.py 1024 2>&1 | tail -10 | tee gpurun_out/r4f/timeline.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; tail -c 1500 gpurun_out/r4f/bench.json; tail -3 gpurun_out/r4f/bench.err
