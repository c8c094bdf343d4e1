#!/bin/bash
# headline / f32 / configs[3] transform timings under build variants of k_xspec (GPU box)
#   tools/run_xspec_variants.sh "" "-DPP_ROW_CHUNK=16" "-DPP_ROW_CHUNK=64"
B="python3 bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 3"
for ex in "$@"; do
  echo "=== EXTRA=$ex"
  make -B -C pulseportraiture_amd/csrc EXTRA="$ex" >/dev/null 2>&1 || { echo build failed; continue; }
  for args in "" "--input-dtype f32" "--workload cfg4-2048x2048-scat --steps 3 --warmup 1"; do
    $B $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  %-28s %-4s %9.1f fits/s  %s' % (d['config']['workload'], d['config'].get('input_dtype',''), d['value'], d['roofline']['all_kernels_ms_per_step']))"
  done
done
make -B -C pulseportraiture_amd/csrc >/dev/null 2>&1
