#!/bin/bash
# GPU box: time the prebuilt libraries under variants/ (tools/build_variants.sh) on configs[1] (512 x 1024):
#   tools/run_prebuilt_variants_cfg2.sh name1 name2 ...     ("base" = the library in the tree)
B="python3 bench.py --no-cpu-baseline --no-other-workloads --workload cfg2-512x1024-phiDM --steps 30 --warmup 5"
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L || continue; fi
  $B 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  %-8s %-24s %9.1f fits/s  %s  checksum %s' % ('$n', d['config']['workload'], d['value'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2]))"
done
done
cp /tmp/lib_orig.so $L
