#!/bin/bash
# What is a wave slot worth to the persistent transform?  The headline (and configs[1]) with the transform's grid at a
# fraction of its residency (PP_GRID_SCALE), alternated on ONE box:   bash tools/run_grid_scale.sh [out]
out=${1:-gpurun_out/r05_grid_scale.txt}
: > $out
for rep in 1 2; do
  for sc in 1.0 0.9375 0.875 0.75 0.5; do
    for wl in toa-4096x2048-phiDM cfg2-512x1024-phiDM; do
      PP_GRID_SCALE=$sc python bench.py --no-cpu-baseline --no-other-workloads --workload $wl --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid x %-7s %-22s %9.0f fits/s %8.3f ms/step  xspec %.3f ms' % ('$sc', '$wl', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['xspec']))" >> $out
    done
  done
done
cat $out
