#!/bin/bash
L=pulseportraiture_amd/csrc/libpptoas_hip.so
cp $L /tmp/lib_orig.so
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/lib_orig.so $L; else cp variants/$n.so $L || continue; fi
  echo "== $n: $(timeout 120 python3 tools/scratch/eval_pass_time.py 2>&1 | grep -E '^eval' )"
done
cp /tmp/lib_orig.so $L
