#!/bin/bash
# GPU box: the reference-seed step (k_rot_mean + k_fps + fit) under prebuilt libraries in variants/
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for n in "$@"; do
  cp variants/$n.so $L || continue
  python3 bench.py --no-cpu-baseline --no-other-workloads --seed-ns -1 --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  %-10s %9.1f fits/s  %s' % ('$n', d['value'], d['roofline']['all_kernels_ms_per_step']))"
done
cp /tmp/lib_orig.so $L
