# Dev tool: PMC passes of the 4-chain-tile HMC kernel (separate rocprofv3 --pmc runs, no trace domains) -> summary JSON.
# Usage (GPU box): bash tools/pmc_hmc_r4.sh
export TMPDIR=/tmp
out=gpurun_out/pmc_r4; rm -rf $out; mkdir -p $out
i=0
# (no FETCH_SIZE / WRITE_SIZE pass: it does not finish for this kernel under rocprofv3 on this pool)
for grp in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python tools/prof_hmc.py 8 > $out/log$i.txt 2>&1
  echo "pass $i ($grp) rc=$?"
done
python tools/pmc_summary.py $out k_hmc_step_r4 3 > $out/summary.json
find $out -name "*.csv" -size +512k -delete
python -c "
import json; d=json.load(open('$out/summary.json'))
for k,v in d.items(): print(k, v if k=='_derived' else v.get('mean_per_launch'))"
