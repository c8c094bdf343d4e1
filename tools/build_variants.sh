#!/bin/bash
# Cross-compile build variants of the library HERE (no GPU needed) into variants/<name>.so;
# tools/run_prebuilt_variants.sh then times them on the GPU box without spending box time on hipcc.
#   tools/build_variants.sh name1 "<EXTRA flags>" name2 "<EXTRA flags>" ...
cd "$(dirname "$0")/../pulseportraiture_amd/csrc" || exit 1
mkdir -p ../../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm -mllvm -amdgpu-atomic-optimizer-strategy=None"
while [ $# -ge 2 ]; do
  n=$1; ex=$2; shift 2
  ( eval hipcc $F $ex -Rpass-analysis=kernel-resource-usage -o ../../variants/$n.so pp_toas.hip > ../../variants/$n.res 2>&1 \
      && echo "built $n" || echo "FAILED $n" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
