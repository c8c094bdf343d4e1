#!/bin/bash
# A/B of option overlap_post (solve + post-fit stage of a deferred batch on the context's second stream, beside the
# next batch's transform) on ONE box, alternated:   bash tools/run_ab_overlap.sh [out-file]
out=${1:-gpurun_out/r05_overlap_ab.txt}
: > $out
for rep in 1 2 3; do
  for wl in toa-4096x2048-phiDM cfg2-512x1024-phiDM cfg3-4096x2048-phiDMGM; do
    for ov in 0 1; do
      python bench.py --no-cpu-baseline --no-other-workloads --workload $wl --steps 30 --warmup 3 --opt overlap_post=$ov 2>/dev/null |
        python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep %-24s overlap_post=$ov %9.0f fits/s %8.3f ms/step kernels %s' % ('$wl', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']))" >> $out
    done
  done
done
for ov in 0 1; do
  python bench.py --no-cpu-baseline --no-other-workloads --seed-ns -1 --steps 30 --warmup 3 --opt overlap_post=$ov 2>/dev/null |
    python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('refseed-in-step overlap_post=$ov %9.0f fits/s %8.3f ms/step kernels %s' % (d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']))" >> $out
done
cat $out
