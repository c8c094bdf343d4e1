#!/bin/bash
# f / gradient rounding of the evaluator and the frequency of SciPy-exit tails under build variants
for ex in "$@"; do
  echo "=== EXTRA=$ex"
  make -B -C pulseportraiture_amd/csrc EXTRA="$ex" >/dev/null 2>&1 || { echo build failed; continue; }
  python tools/dev_grad_noise.py 2>&1 | grep "f err"
  python tools/dev_scat_tails.py 2>&1 | grep "device\|dphi\|True\|False"
done
make -B -C pulseportraiture_amd/csrc >/dev/null 2>&1
