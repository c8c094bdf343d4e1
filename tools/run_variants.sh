#!/bin/bash
# usage: run_variants.sh "<EXTRA flags>" ...   (scratch helper for kernel experiments on the GPU box:
# rebuilds the library with each set of flags and prints the headline's kernel times)
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['input_dtype'], d['value'], d['roofline']['all_kernels_ms_per_step'], d['convergence']['nfeval_mean'])"; }
for ex in "$@"; do
  echo "=== EXTRA=$ex"
  make -B -C pulseportraiture_amd/csrc EXTRA="$ex" >/dev/null 2>&1 || { echo build failed; continue; }
  python bench.py --no-cpu-baseline --no-other-workloads --truth-guesses --steps 4 | pj
  python bench.py --no-cpu-baseline --no-other-workloads --truth-guesses --steps 4 --input-dtype f32 | pj
done
make -B -C pulseportraiture_amd/csrc >/dev/null 2>&1
