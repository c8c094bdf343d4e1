#!/bin/bash
# configs[3] bench lines with the scattering model of the closing iterations on / off
B="python3 bench.py --no-cpu-baseline --no-other-workloads --workload cfg4-2048x2048-scat --steps 3 --warmup 1"
for m in trust-ncg newton; do
  for sm in 0 1; do
    echo "== $m scat_model=$sm"
    $B --method $m --opt scat_model=$sm 2>/dev/null | tail -1
  done
done
