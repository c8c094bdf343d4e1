#!/bin/bash
# A/B of PP_SLOT_EARLY_EXIT on ONE box, alternated: the library in the tree ("base": the slot loop ends with the last kept
# slot) against variants/noexit.so (-DPP_SLOT_EARLY_EXIT=0: every slot walked).  Checksums must be identical.
#   tools/build_variants.sh noexit "-DPP_SLOT_EARLY_EXIT=0";  bash tools/run_ab_slot_exit.sh [out]
out=${1:-gpurun_out/r05_slot_exit_ab.txt}
: > $out
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
run() {
  python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 3 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-34s %9.0f fits/s %8.3f ms/step  xspec %.3f  checksum %s' % ('$1', '$2' or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['xspec'], d['gathered_records']['checksum'][:2]))" >> $out
}
for rep in 1 2 3; do
  for n in base noexit; do
    if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L; fi
    run $n ""
    run $n "--input-dtype f32"
    run $n "--workload cfg2-512x1024-phiDM"
    run $n "--workload cfg4-2048x2048-scat"
    run $n "--variant measured_noise"
  done
done
cp /tmp/lib_orig.so $L
cat $out
