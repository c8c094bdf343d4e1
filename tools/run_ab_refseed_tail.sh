#!/bin/bash
# A/B of the fused tail for get_TOAs' default flow (the reference's guess inside the pass, k_xspec_qr1024) on ONE box,
# alternated three times: fuse_tail=0 (the guess's finish, fit_phase_shift, start points, solve and post-fit stage by
# stand-alone kernels behind the pass) against fuse_tail=1 (all of it as tickets of the NEXT batch's pass), and the
# round-5 library (variants/base.so, PP_TOAS_LIB) beside them:   bash tools/run_ab_refseed_tail.sh [out]
out=${1:-gpurun_out/r06_refseed_tail_ab.txt}
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s %-34s %9.0f fits/s %8.3f ms/step  %s  checksum %s' % (sys.argv[1], sys.argv[2] or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['gathered_records']['checksum'][:2]))" "$1" "$2" >> $out; }
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 30 --warmup 3 --pipeline 3"
for rep in 1 2 3; do
  for wl in "--seed-ns -1" "--seed-ns -1 --variant masked20" "--seed-ns -1 --input-dtype f32" ""; do
    [ -f variants/base.so ] && PP_TOAS_LIB=variants/base.so $B $wl 2>/dev/null | line "round5 lib" "$wl"
    for ft in 0 1; do
      $B $wl --opt fuse_tail=$ft 2>/dev/null | line "fuse_tail=$ft" "$wl"
    done
  done
done
cat $out
