# Round-5 measurement session (one gpurun call): GPU suite, bench line + rocprofv3 kernel stats of the same command, HBM traffic
# and issue counters of the kernels the roofline rows name (each PMC pass its own run), secondary workloads.
set -x
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/final_r5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json      # first: what the driver's round-end run sees (a fresh box)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
bash tools/pmc_traffic.sh k_hmc_step_r4 $O/pmc_traffic_r4 3 > $O/pmc_traffic_r4.log 2>&1; cp $O/pmc_traffic_r4/summary.json $O/hmc_step_r4_traffic_pmc_summary.json
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_cfg4_1gpu.json
FABHIP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_gloo.json
FABHIP_SHARDED_ONE_OP=0 FABHIP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_gloo_python_loop.json
[ -x tools/ubench/bin/xcu ] && timeout 120 tools/ubench/bin/xcu > $O/ubench_xcu.txt 2>&1
python tools/time_hmc_shapes.py 2>/dev/null | grep "W=" > $O/hmc_tile_shapes.txt
python tools/timeline_r8.py 2048 2>/dev/null | tail -11 > $O/hmc_r8f_stage_timeline.txt
(echo "--- unfused (FABHIP_R4_STREAM=1) ---"; FABHIP_R4_STREAM=1 python tools/timeline_r8.py 2048 2>/dev/null | tail -13) >> $O/hmc_r8f_stage_timeline.txt
timeout 300 python tools/bench_spline.py 2>/dev/null | tail -1 > $O/spline_cfg3.json
timeout 300 python tools/timeline_spline.py 2>/dev/null | tail -13 > $O/spline_r8_stage_timeline.txt
CFG=5 N=3 timeout 300 python tools/bench_spline.py 2>/dev/null | tail -1 > $O/spline_cfg5_shape.json
bash tools/pmc_stream_kernels.sh > $O/pmc_stream.log 2>&1
cp gpurun_out/pmc_stream_hmc/summary.json $O/hmc_step_r8_pmc_summary.json; cp gpurun_out/pmc_stream_spline/summary.json $O/spline_r8_pmc_summary.json
rm -rf $O/pmc_traffic_r4 gpurun_out/pmc_stream_hmc gpurun_out/pmc_stream_spline
bash tools/trace_step.sh > /dev/null 2>&1; cp gpurun_out/trace_step/step_timeline.txt $O/step_timeline.txt
timeout 300 python tools/host_overhead.py 2>/dev/null | tail -9 > $O/host_overhead.txt
timeout 300 python tools/timeline_r4.py 1024 2>/dev/null | tail -10 > $O/hmc_r4f_stage_timeline.txt
FABHIP_R4_STREAM=1 timeout 300 python tools/timeline_r4.py 1024 2>/dev/null | tail -14 > $O/hmc_r4s_stage_timeline.txt
(echo "LDS stash (FABHIP_R4_STREAM=3, three items of each W x W stage through LDS), stage stamps in cycles"; FABHIP_R4_STREAM=3 timeout 300 python tools/timeline_r4.py 1024 2>/dev/null | tail -10; echo "--- ring alone (default) ---"; cat $O/hmc_r4f_stage_timeline.txt) > $O/hmc_r4f_stash_timeline.txt
(echo -n "stash: "; FABHIP_R4_STREAM=3 timeout 300 python tools/time_hmc.py 1024 2>/dev/null | tail -1; echo -n "ring alone: "; timeout 300 python tools/time_hmc.py 1024 2>/dev/null | tail -1) >> $O/hmc_r4f_stash_timeline.txt
timeout 300 python tools/time_hmc.py 1024 2>/dev/null | tail -1 > $O/hmc_r4_fused_vs_stream.txt
FABHIP_R4_STREAM=1 timeout 300 python tools/time_hmc.py 1024 2>/dev/null | tail -1 >> $O/hmc_r4_fused_vs_stream.txt
timeout 600 python tools/bench_multinomial.py > $O/multinomial.json 2>/dev/null
PROF_SCRIPT=tools/prof_resample.py bash tools/pmc_traffic.sh k_sample_multinomial $O/pmc_multinomial > $O/pmc_multinomial.log 2>&1; cp $O/pmc_multinomial/summary.json $O/sample_multinomial_pmc_summary.json; rm -rf $O/pmc_multinomial
REPS=10 timeout 900 python tools/soak_stream_kernels.py 2>&1 | tail -12 > $O/soak.txt
FABHIP_TEST_REPORT=$PWD/$O/trainer_replay_outliers.txt timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep "passed\|failed\|FAILED" > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
ls -la $O
