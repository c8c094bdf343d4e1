#!/bin/bash
# GPU box: the small kernels behind the transform (taylor_solve, finalize) under build variants, on the headline
# and configs[1]:   tools/run_prebuilt_variants_small.sh base name1 ...
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L || continue; fi
  for w in "toa-4096x2048-phiDM --steps 6 --warmup 2" "cfg2-512x1024-phiDM --steps 30 --warmup 5"; do
    python3 bench.py --no-cpu-baseline --no-other-workloads --workload $w 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']
print('  %-6s %-22s %9.1f fits/s  taylor_solve %.4f finalize %.4f xspec %.3f' % ('$n', d['config']['workload'], d['value'], k.get('taylor_solve',0), k.get('finalize',0), k.get('xspec',0)))"
  done
done
done
cp /tmp/lib_orig.so $L
