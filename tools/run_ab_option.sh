#!/bin/bash
# GPU box: headline / f32 timings with an engine option off and on.  tools/run_ab_option.sh one_exchange
B="python3 bench.py --no-cpu-baseline --no-other-workloads --steps ${PP_AB_STEPS:-10} --warmup 3 $PP_AB_ARGS"
for val in 0 1 0 1; do
  for args in "" "--input-dtype f32"; do
    $B $args --opt $1=$val 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  $1=$val %-22s %-4s %9.1f fits/s  %s  checksum %s nfev %s' % (d['config']['workload'], d['config'].get('input_dtype',''), d['value'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2], d['convergence']['nfeval_mean']))"
  done
done
