# Round-6 measurement session (one gpurun call): bench line + rocprofv3 kernel stats of the same command, the trainer iteration
# (tools/bench_trainer.py) with its rocprofv3 kernel stats and launch timeline, HBM traffic of the kernels the roofline rows
# name (each PMC pass its own run), secondary workloads, the GPU suite.
set -x
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/final_r6; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json      # first: what the driver's round-end run sees (a fresh box)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
bash tools/trace_trainer.sh > /dev/null 2>&1
cp gpurun_out/trace_trainer/kernel_stats.csv $O/trainer_kernel_stats_rocprofv3.csv
cp gpurun_out/trace_trainer/iteration_timeline.txt $O/trainer_iteration_timeline.txt
cp gpurun_out/trace_trainer/bench.json $O/trainer_under_rocprof.json
timeout 600 python tools/bench_trainer.py 2>/dev/null | tail -1 > $O/trainer_iteration.json
bash tools/pmc_traffic.sh k_hmc_step_r4 $O/pmc_traffic_r4 3 > $O/pmc_traffic_r4.log 2>&1; cp $O/pmc_traffic_r4/summary.json $O/hmc_step_r4_traffic_pmc_summary.json
ITERS=2 PROF_SCRIPT=tools/bench_trainer.py bash tools/pmc_traffic.sh k_flow_log_prob_tape_r8 $O/pmc_tape > $O/pmc_tape.log 2>&1; cp $O/pmc_tape/summary.json $O/tape_r8_traffic_pmc_summary.json
ITERS=2 PROF_SCRIPT=tools/bench_trainer.py bash tools/pmc_traffic.sh k_pgrad_tiles $O/pmc_pgrad > $O/pmc_pgrad.log 2>&1; cp $O/pmc_pgrad/summary.json $O/pgrad_tiles_traffic_pmc_summary.json
rm -rf $O/pmc_traffic_r4 $O/pmc_tape $O/pmc_pgrad
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_cfg4_1gpu.json
FABHIP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_gloo.json
python tools/time_hmc_shapes.py 2>/dev/null | grep "W=" > $O/hmc_tile_shapes.txt
python tools/timeline_r8.py 2048 2>/dev/null | tail -11 > $O/hmc_r8f_stage_timeline.txt
timeout 300 python tools/bench_spline.py 2>/dev/null | tail -1 > $O/spline_cfg3.json
timeout 300 python tools/timeline_spline.py 2>/dev/null | tail -13 > $O/spline_r8_stage_timeline.txt
bash tools/trace_step.sh > /dev/null 2>&1; cp gpurun_out/trace_step/step_timeline.txt $O/step_timeline.txt
timeout 300 python tools/host_overhead.py 2>/dev/null | tail -9 > $O/host_overhead.txt
timeout 300 python tools/timeline_r4.py 1024 2>/dev/null | tail -10 > $O/hmc_r4f_stage_timeline.txt
timeout 600 python tools/bench_multinomial.py > $O/multinomial.json 2>/dev/null
REPS=10 timeout 900 python tools/soak_stream_kernels.py 2>&1 | tail -12 > $O/soak.txt
FABHIP_TEST_REPORT=$PWD/$O/trainer_replay_outliers.txt timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep "passed\|failed\|FAILED" > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
ls -la $O
