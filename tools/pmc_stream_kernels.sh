# Dev tool: PMC passes (MFMA ops / busy cycles, waits, LDS conflicts, L2 hits) of the two 4x4x1 stream kernels:
# k_hmc_step_r8 at 2048 chains (tools/prof_hmc.py) and k_spline_logprob_r8 at cfg 3's shape (tools/prof_spline.py).
# Separate rocprofv3 --pmc runs restricted to the kernel, no trace domains.  Usage (GPU box): bash tools/pmc_stream_kernels.sh
export TMPDIR=/tmp
for which in hmc spline; do
  if [ $which = hmc ]; then kern=k_hmc_step_r8; script="tools/prof_hmc.py 4 2048"; else kern=k_spline_logprob_r8; script="tools/prof_spline.py 4 2048"; fi
  out=gpurun_out/pmc_stream_$which; rm -rf $out; mkdir -p $out
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $grp --kernel-include-regex "$kern" --output-format csv -d $out/p$i -- python $script > $out/log$i.txt 2>&1
    echo "$which pass $i ($grp) rc=$?"
  done
  python tools/pmc_summary.py $out $kern 1 > $out/summary.json
  find $out -name "*.csv" -size +512k -delete
  python -c "
import json; d=json.load(open('$out/summary.json'))
for k,v in d.items(): print(k, v if k=='_derived' else v.get('mean_per_launch'))"
done
