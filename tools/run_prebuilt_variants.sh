#!/bin/bash
# GPU box: time the prebuilt libraries under variants/ (tools/build_variants.sh) on the headline,
# f32 and configs[3] workloads.   tools/run_prebuilt_variants.sh name1 name2 ...
B="python3 bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 3"
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for n in "$@"; do
  echo "=== $n"
  if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L || continue; fi
  for args in "" "--input-dtype f32" ${PP_VARIANT_CFG4:+"--workload cfg4-2048x2048-scat --steps 3 --warmup 1"}; do
    $B $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  %-28s %-4s %9.1f fits/s  %s  parity-checksum %s' % (d['config']['workload'], d['config'].get('input_dtype',''), d['value'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2]))"
  done
done
cp /tmp/lib_orig.so $L
