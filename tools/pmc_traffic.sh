# Dev tool: HBM-side traffic of one kernel (FETCH_SIZE / WRITE_SIZE and their raw TCC_EA0 components), one rocprofv3
# --pmc pass per counter group, restricted to the kernel by name, few launches, each pass bounded by its own timeout.
# Usage (GPU box): bash tools/pmc_traffic.sh <kernel-regex> <out-dir> [prof_hmc.py args...]
export TMPDIR=/tmp
kern=${1:-k_hmc_step_r4}; out=${2:-gpurun_out/pmc_traffic}; shift 2
rm -rf $out; mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  t0=$(date +%s)
  timeout 150 rocprofv3 --pmc $grp --kernel-include-regex "$kern" --output-format csv -d $out/p$i -- python ${PROF_SCRIPT:-tools/prof_hmc.py} "$@" > $out/log$i.txt 2>&1
  echo "pass $i ($grp) rc=$? $(( $(date +%s) - t0 )) s"
done
python tools/pmc_summary.py $out "$kern" 1 > $out/summary.json
find $out -name "*.csv" -size +512k -delete
python -c "
import json; d=json.load(open('$out/summary.json'))
for k,v in d.items(): print(k, v if k=='_derived' else v)"
