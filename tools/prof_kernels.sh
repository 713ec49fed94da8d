# usage: tools/prof_kernels.sh "<python command>" <grep pattern>   - rocprofv3 kernel stats of a command, rows matching the pattern
export TMPDIR=/tmp; rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- $1 > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
python - "$f" "$2" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
