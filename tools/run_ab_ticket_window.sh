#!/bin/bash
# A/B of the window over which the waves ask for their tickets (PP_TICKET_WINDOW_NUM / DEN of a wave's share; product 3/4):
out=${1:-gpurun_out/r06_ticket_window_ab.txt}
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-34s %9.0f fits/s %8.3f ms/step  xspec %.3f  checksum %s' % (sys.argv[1], sys.argv[2] or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['xspec'], d['gathered_records']['checksum'][:2]))" "$1" "$2" >> $out; }
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 30 --warmup 3 --pipeline 3"
for rep in 1 2; do
  for wl in "" "--variant masked20" "--input-dtype f32"; do
    $B $wl 2>/dev/null | line "3/4 product" "$wl"
    for v in win14 win12 win11 win32; do
      PP_TOAS_LIB=variants/$v.so $B $wl 2>/dev/null | line "$v" "$wl"
    done
  done
done
cat $out
