#!/bin/bash
# First N > 1 session on a multi-GPU MI355X node: bench.py --gpus {1,2,4,8} x {headline, cfg4}, one table.
#   bash tools/scale_run.sh [out_dir]         (from the repo root; needs as many GPUs as the largest N)
# Every run is bounded (timeout + bench.py's process-group watchdog, FABHIP_BENCH_PG_TIMEOUT): a hang is an error row,
# not a stuck lease.  The tuned row (`value`) and the eval-mode row (`value_eval_mode`, step sizes frozen: no acceptance-slab
# all-gathers) come from the same run; bench.py itself checks rccl_ranks == N, collectives_per_step == M + 1 (tuned) / 1 (eval).
# Scaling efficiency is NOT computed here from a shared GPU: rows for N above the node's GPU count are skipped.
set -u
OUT=${1:-gpurun_out/scale}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=1 FABHIP_BENCH_PG_TIMEOUT=${FABHIP_BENCH_PG_TIMEOUT:-180}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "GPUs on this node: $NGPU" | tee "$OUT/table.txt"
printf "%-9s %3s %14s %14s %10s %6s %6s %s\n" workload N samples/s eval_samples/s ms/step coll rccl status | tee -a "$OUT/table.txt"
for WL in headline cfg4; do
  for N in 1 2 4 8; do
    if [ "$N" -gt "$NGPU" ]; then
      printf "%-9s %3d %14s %14s %10s %6s %6s %s\n" $WL $N - - - - - "skipped (node has $NGPU GPUs)" | tee -a "$OUT/table.txt"
      continue
    fi
    LOG="$OUT/bench_${WL}_n${N}"
    timeout -k 10 900 python bench.py --gpus $N --workload $WL --steps 30 --warmup 10 --no-cpu-baseline > "$LOG.json" 2> "$LOG.err"
    RC=$?
    python - "$LOG.json" "$WL" "$N" "$RC" <<'PY' | tee -a "$OUT/table.txt"
import json, sys
path, wl, n, rc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
row = None
try:
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith("{"):
            row = json.loads(ln)
except OSError:
    pass
if row and "value" in row:
    st = "ok" if (rc == 0 and row.get("multi_gpu_checks") == "ok") else f"rc={rc} checks={row.get('multi_gpu_checks')}"
    print("%-9s %3d %14.1f %14.1f %10.3f %6s %6s %s" % (wl, n, row["value"], row["value_eval_mode"], row["ms_per_step"],
                                                      row.get("collectives_per_step"), row.get("rccl_ranks"), st))
else:
    why = (row or {}).get("error", "no JSON line (timeout 124 = hang; see the .err file)")
    print("%-9s %3d %14s %14s %10s %6s %6s rc=%d %s" % (wl, n, "-", "-", "-", "-", "-", rc, why))
PY
  done
done
