# Dev tool: the measurements committed under profiles/ in one gpurun call (bench line, rocprofv3 kernel stats of the same
# command, secondary configs / resample kernels, then the GPU suite).  Usage (GPU box): bash tools/final_run.sh
set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/final
python bench.py 2>/dev/null | tail -1 > gpurun_out/final/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/final/bench_prof.log 2>&1
find gpurun_out/final/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/final/kernel_stats.csv
rm -rf gpurun_out/final/prof
python tools/bench_extra.py > gpurun_out/final/bench_extra.json 2> gpurun_out/final/bench_extra.err
python tools/bench_multinomial.py > gpurun_out/final/bench_multinomial.json 2>/dev/null

# spline flow (cfg 3 / cfg 5 shapes): AIS rates, stage timeline of the 8-chain kernel, per-kernel times, stream micro-benchmark
python tools/bench_spline.py 2>/dev/null | tail -1 > gpurun_out/final/spline_cfg3.json
FABHIP_SPLINE_MFMA=16 python tools/bench_spline.py 2>/dev/null | tail -1 > gpurun_out/final/spline_cfg3_16x16x4.json
CFG=5 N=3 python tools/bench_spline.py 2>/dev/null | tail -1 > gpurun_out/final/spline_cfg5_shape.json
python tools/timeline_spline.py 2>/dev/null | tail -14 > gpurun_out/final/spline_r8_stage_timeline.txt
FABHIP_TILE=16 python tools/timeline_spline.py 2>/dev/null | tail -14 > gpurun_out/final/spline_r8_16chain_stage_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_sp -- python tools/bench_spline.py > gpurun_out/final/spline_prof.log 2>&1
find gpurun_out/final/prof_sp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/final/spline_cfg3_kernel_stats.csv
rm -rf gpurun_out/final/prof_sp
[ -x tools/ubench/bin/nsplit ] && tools/ubench/bin/nsplit > gpurun_out/final/ubench_nsplit.txt 2>&1
python tools/time_hmc_shapes.py 2>/dev/null | grep "W=" > gpurun_out/final/hmc_tile_shapes.txt
python tools/timeline_r8.py 2048 2>/dev/null | tail -13 > gpurun_out/final/hmc_r8_stage_timeline.txt
python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/final/bench_cfg4_1gpu.json
FABHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/bench_2ranks_one_gpu_gloo.json
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
