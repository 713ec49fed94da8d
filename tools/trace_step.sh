cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/trace_step; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --steps 6 --warmup 3 > $O/log.txt 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $O/step_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find k_ais_init_r4 occurrences
idx = [i for i, r in enumerate(rows) if "k_ais_init_r4" in r["Kernel_Name"]]
a, b = idx[5], idx[6]
t0 = int(rows[a - 1]["End_Timestamp"])
prev_end = None
for r in rows[a - 6:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
    prev_end = e
PY
rm -rf $O/prof
cat $O/step_timeline.txt
