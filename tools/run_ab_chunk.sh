#!/bin/bash
# A/B of the evaluators' channel-chunk size (PP_CHUNK_CHANNELS: 256 product / 64 variants/chunk64.so) on ONE box, alternated:
#   bash tools/run_ab_chunk.sh [out]
out=${1:-gpurun_out/r06_chunk_ab.txt}
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-44s %9.0f fits/s %8.3f ms/step  %s' % (sys.argv[1], sys.argv[2] or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']))" "$1" "$2" >> $out; }
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 3"
for rep in 1 2 3; do
  for wl in "--workload cfg4-2048x2048-scat" "--workload cfg4-2048x2048-scat --method newton" "--workload cfg4-2048x2048-scat --seed-ns -1" "--seed-ns 100"; do
    $B $wl 2>/dev/null | line "chunk 256" "$wl"
    PP_TOAS_LIB=variants/chunk64.so $B $wl 2>/dev/null | line "chunk 64" "$wl"
  done
done
cat $out
