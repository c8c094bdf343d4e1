#!/bin/bash
# A/B/C... of the library in the tree ("base") against several variants/<name>.so on ONE box, alternated three times:
#   bash tools/run_ab_variants.sh "<name1> <name2> ..." <out-file> ["bench args" ...]
vs=$1; out=$2; shift 2
[ $# -eq 0 ] && set -- ""
: > $out
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2 3; do
  for n in base $vs; do
    if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L; fi
    for args in "$@"; do
      python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 3 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-34s %9.0f fits/s %8.3f ms/step  xspec %.3f  checksum %s' % ('$n', '$args' or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'].get('xspec', 0), d['gathered_records']['checksum'][:2]))" >> $out
    done
  done
done
cp /tmp/lib_orig.so $L
cat $out
