#!/bin/bash
# on the GPU box: time one HMC transition (+ stage timeline) with each stashed library
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/price
for tag in "$@"; do
  cp tools/experiments/price/$tag/libfabhip.so tools/experiments/price/$tag/_fabhip_torch.so tools/experiments/price/$tag/libfabhip.so.srchash fab_torch_amd/
  export FABHIP_EXTRA_FLAGS="$(cat tools/experiments/price/$tag/flags)"
  echo "== $tag ($FABHIP_EXTRA_FLAGS)" | tee -a gpurun_out/price/out.txt
  FABHIP_SKIP_ISA_CHECK=1 timeout 600 python tools/time_hmc.py 1024 2>&1 | tail -1 | tee -a gpurun_out/price/out.txt
  FABHIP_SKIP_ISA_CHECK=1 timeout 600 python tools/timeline_r4.py 1024 2>&1 | tail -16 | tee -a gpurun_out/price/out.txt
done
