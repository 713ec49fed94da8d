cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_hmc_shapes.py -x -q -k "fused_stage or four_chain" 2>&1 | tail -15 | tee gpurun_out/r4f/pytest1.txt
timeout 300 python tools/time_hmc.py 1024 2>&1 | tail -2 | tee gpurun_out/r4f/time.txt
timeout 300 python tools/timeline_r4.py 1024 2>&1 | tail -12 | tee gpurun_out/r4f/timeline.txt
FABHIP_R4_STREAM=1 timeout 300 python tools/time_hmc.py 1024 2>&1 | tail -2 | tee -a gpurun_out/r4f/time.txt
