#!/bin/bash
# A/B of the degree-4 re-expansion of the scattering model's closing rounds (option scat_model_reexp) on ONE box, alternated:
#   bash tools/run_ab_reexp.sh [out]
out=${1:-gpurun_out/r06_reexp_ab.txt}
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['convergence']
print('%-22s %-44s %9.0f fits/s %8.3f ms/step  %s  nfeval %.3f  checksum %s' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], c.get('nfeval_mean',0), d['gathered_records']['checksum'][:2]))" "$1" "$2" >> $out; }
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 3"
for rep in 1 2; do
  for wl in "--workload cfg4-2048x2048-scat" "--workload cfg4-2048x2048-scat --seed-ns -1"; do
    for v in 0 1; do
      $B $wl --opt scat_model_reexp=$v 2>/dev/null | line "scat_model_reexp=$v" "$wl"
    done
  done
done
cat $out
