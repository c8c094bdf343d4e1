#!/bin/bash
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for n in "$@"; do
  cp variants/$n.so $L || continue
  for m in trust-ncg newton; do
  python3 bench.py --workload cfg4-2048x2048-scat --method $m --no-cpu-baseline --no-other-workloads --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  %-6s %-9s %9.1f fits/s  %s' % ('$n', '$m', d['value'], d['roofline']['all_kernels_ms_per_step']))"
  done
done
cp /tmp/lib_orig.so $L
