for m in 0 8 0 8; do FABHIP_TAPE_TILES=$m bash tools/trace_trainer.sh > /dev/null 2>&1; echo "TAPE_TILES=$m $(tail -1 gpurun_out/trace_trainer/iteration_timeline.txt)"; python - <<'PY'
import re
rows=[l for l in open("gpurun_out/trace_trainer/iteration_timeline.txt") if "us" in l and not l.startswith("#")]
import collections
d=collections.defaultdict(list)
for l in rows:
    m=re.match(r"\s*([-\d.]+) us\s+\+\s*([-\d.]+) gap\s+([\d.]+) us\s+(.*)", l)
    if m: d[m.group(4)[:40]].append(float(m.group(3)))
for k,v in d.items():
    if len(v)>=7 and ("tape" in k or "minibatch" in k or "pgrad" in k or "adam" in k or "sqnorm" in k): print("   %-42s n=%d avg %.1f" % (k, len(v), sum(v)/len(v)))
PY
done
