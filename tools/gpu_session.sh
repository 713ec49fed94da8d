#!/bin/bash
# One GPU-box session producing the evidence DESIGN.md / profiles/ cite.  Run through gpurun from the repo root:
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh r1'
# Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/<round>/.
set -u
TAG=${1:-r1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC_SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
          "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
          "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")

if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
  tail -3 "$OUT/pytest_gpu.log"
fi

timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"

# kernel trace + stats of the same command (CPU baseline leg skipped: it launches no kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python bench.py --no-cpu-baseline > "$OUT/bench_traced.json" 2> "$OUT/trace.err"

# PMC passes, each in its own run (never combined with a trace domain)
i=0
for set in "${PMC_SETS[@]}"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc/pass$i" -o hmc -- \
      python tools/prof_hmc.py 6 > "$OUT/pmc_pass$i.log" 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py "$OUT/pmc" k_hmc_step 1 > "$OUT/hmc_step_pmc_summary.json" 2>> "$OUT/pmc_summary.err"

if [ "${SKIP_EXTRA:-0}" != 1 ]; then
  timeout 600 python tools/bench_extra.py > "$OUT/bench_extra.json" 2> "$OUT/bench_extra.err"
  timeout 600 python tools/prof_train.py > "$OUT/prof_train.json" 2> "$OUT/prof_train.err"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_resample" -o resample -- \
      python tools/prof_resample.py > "$OUT/trace_resample.log" 2>&1
  cat "$OUT/prof_train.json"
fi
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
