#!/bin/bash
# A/B of option fuse_tail on ONE box, alternated:   bash tools/run_ab_fuse_tail.sh [out]
out=${1:-gpurun_out/r05_fuse_tail_ab.txt}
: > $out
for rep in 1 2 3; do
  for wl in "" "--workload cfg2-512x1024-phiDM" "--workload cfg3-4096x2048-phiDMGM" "--variant masked20" "--input-dtype f32"; do
    for cfg in "0 2" "0 3" "1 3"; do
      set -- $cfg
      python bench.py --no-cpu-baseline --no-other-workloads $wl --steps 30 --warmup 3 --pipeline $2 --opt fuse_tail=$1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s fuse_tail=%s in flight %s %9.0f fits/s %8.3f ms/step  %s  checksum %s' % ('$wl' or 'headline', '$1', '$2', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2]))" >> $out
    done
  done
done
cat $out
