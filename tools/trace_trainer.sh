# rocprofv3 kernel trace + stats of tools/bench_trainer.py: per-kernel summary and the launch timeline of ONE trainer iteration
# (from one chain-initialisation kernel to the next).  Output: gpurun_out/trace_trainer/{kernel_stats.csv, iteration_timeline.txt, bench.json}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/trace_trainer; mkdir -p $O
ITERS=${ITERS:-8} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/bench_trainer.py > $O/log.txt 2>&1
grep "^{\"workload\"" $O/log.txt | tail -1 > $O/bench.json
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $O/iteration_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_ais_init" in r["Kernel_Name"]]
# the trainer steps come first (warm-up + timed), the AIS-only calls after: take an iteration from the middle of the timed steps
gaps = [idx[i + 1] - idx[i] for i in range(len(idx) - 1)]
long = [i for i, g in enumerate(gaps) if g > 40]
a = idx[long[len(long) // 2]]; b = idx[long[len(long) // 2] + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:110]}")
    prev_end = e
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"# iteration span {span / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, {b - a} launches")
PY
rm -rf $O/prof
tail -3 $O/iteration_timeline.txt; cat $O/bench.json
