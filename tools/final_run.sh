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
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
