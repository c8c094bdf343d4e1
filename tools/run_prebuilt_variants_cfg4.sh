#!/bin/bash
# GPU box: time the prebuilt libraries under variants/ (tools/build_variants.sh) on configs[3]
# (2048 x 2048 scattering fit), both methods, two rounds.  tools/run_prebuilt_variants_cfg4.sh base name1 ...
B="python3 bench.py --workload cfg4-2048x2048-scat --no-cpu-baseline --no-other-workloads --steps ${PP_AB_STEPS:-8} --warmup 2"
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L || continue; fi
  for m in trust-ncg newton; do
    $B --method $m 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  %-14s %-10s %9.1f fits/s  %s  checksum %s' % ('$n', '$m', d['value'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2]))"
  done
done
done
cp /tmp/lib_orig.so $L
