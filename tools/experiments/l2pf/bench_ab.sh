#!/bin/bash
# on the GPU box: the headline bench line (value, ms per transition launch) with each stashed library
cd "$(dirname "$0")/../../.."
for tag in "$@"; do
  cp tools/experiments/l2pf/$tag/libfabhip.so tools/experiments/l2pf/$tag/_fabhip_torch.so tools/experiments/l2pf/$tag/libfabhip.so.srchash fab_torch_amd/
  export FABHIP_EXTRA_FLAGS="$(cat tools/experiments/l2pf/$tag/flags)"
  echo -n "== $tag: "
  FABHIP_SKIP_ISA_CHECK=1 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.0f  ms/step %.4f  ms/launch %.4f  frac %.4f  2048: %.4f  4096: %.4f  fast %.0f  spline %.0f' % (d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'], r['chains_2048']['ms_per_launch'], r['full_chip']['ms_per_launch'], d['fast_mode']['value'], d['spline_cfg3']['value']))"
done
