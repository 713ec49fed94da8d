# Dev tool: what bounds k_emit_systematic (N = 2^26) - instruction mix and busy cycles from rocprofv3 --pmc, one pass per
# counter group restricted to the kernel, + HBM traffic; summarised by tools/pmc_summary.py.  Usage (GPU box): bash tools/pmc_emit.sh <out-dir>
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_emit}
rm -rf $out; mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-include-regex "k_emit_systematic" --output-format csv -d $out/p$i -- python tools/prof_systematic.py > $out/log$i.txt 2>&1
  echo "pass $i ($grp) rc=$?"
done
python tools/pmc_summary.py $out k_emit_systematic 1 > $out/summary.json
find $out -name "*.csv" -size +512k -delete
cat $out/summary.json
