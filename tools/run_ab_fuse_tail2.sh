#!/bin/bash
# A/B of option fuse_tail (the previous enqueued batch's solve + post-fit stage as tickets of this batch's transform)
# on ONE box, alternated, steps three deep either way:   bash tools/run_ab_fuse_tail2.sh [out]
out=${1:-gpurun_out/r05_fuse_tail_ab2.txt}
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-36s %9.0f fits/s %8.3f ms/step  %s  checksum %s' % (sys.argv[1], sys.argv[2] or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2]))" "$1" "$2" >> $out; }
for rep in 1 2 3; do
  for wl in "" "--workload cfg3-4096x2048-phiDMGM" "--variant masked20" "--input-dtype f32" "--variant measured_noise" "--workload cfg2-512x1024-phiDM"; do
    for ft in 0 1; do
      python bench.py --no-cpu-baseline --no-other-workloads $wl --steps 30 --warmup 3 --pipeline 3 --opt fuse_tail=$ft 2>/dev/null | line "fuse_tail=$ft" "$wl"
    done
  done
done
cat $out
