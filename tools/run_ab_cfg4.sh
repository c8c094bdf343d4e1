#!/bin/bash
# GPU box: configs[3] (2048 x 2048 scattering fit) with an engine option at several values, both
# methods.  tools/run_ab_cfg4.sh eval_lpc 0 8 16
opt=$1; shift
B="python3 bench.py --workload cfg4-2048x2048-scat --no-cpu-baseline --no-other-workloads --steps ${PP_AB_STEPS:-8} --warmup 2 $PP_AB_ARGS"
for rep in 1 2; do
  for val in "$@"; do
    for m in trust-ncg newton; do
      $B --method $m --opt $opt=$val 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  $opt=$val %-10s %9.1f fits/s  %s  checksum %s nfev %s' % ('$m', d['value'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2], d['convergence']['nfeval_mean']))"
    done
  done
done
