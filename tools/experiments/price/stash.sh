#!/bin/bash
# build the library with extra compile flags ($2..) and stash it under tools/experiments/price/<tag $1> (timing experiments)
set -e
tag=$1; shift
cd "$(dirname "$0")/../../.."
export FABHIP_EXTRA_FLAGS="$*"
python -c "from fab_torch_amd import _build; _build.build(force=False)" 2>&1 | grep -v remark | tail -3
mkdir -p tools/experiments/price/$tag
cp fab_torch_amd/libfabhip.so fab_torch_amd/_fabhip_torch.so fab_torch_amd/libfabhip.so.srchash tools/experiments/price/$tag/
echo "$*" > tools/experiments/price/$tag/flags
