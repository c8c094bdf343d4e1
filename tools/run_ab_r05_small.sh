#!/bin/bash
# round-5 A/Bs of host-side knobs on ONE box, alternated:   bash tools/run_ab_r05_small.sh [out-file]
#   configs[3] trust-ncg: when the host first looks at the count of unfinished subints (check_from) and how often (check_every)
#   reference-seed flow: channel stride of its pilot pass (refseed_stride)
out=${1:-gpurun_out/r05_small_ab.txt}
: > $out
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['all_kernels_ms_per_step']
print('%-44s %9.0f fits/s %8.3f ms/step  outside kernels %.3f  nfeval %.2f npass %.2f  %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step']-sum(k.values()), d['convergence']['nfeval_mean'], d['convergence']['npass_mean'], k))" "$1" >> $out
}
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 3"
for rep in 1 2; do
  for opt in "check_from=2" "check_from=4" "check_from=5" "check_from=6" "check_from=4 --opt check_every=2" "check_from=5 --opt check_every=2"; do
    $B --workload cfg4-2048x2048-scat --opt $opt 2>/dev/null | line "cfg4 trust-ncg $opt"
  done
  for opt in "refseed_stride=64" "refseed_stride=128"; do
    $B --seed-ns -1 --opt $opt 2>/dev/null | line "reference seed in step $opt"
  done
done
cat $out
