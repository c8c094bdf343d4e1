#!/bin/bash
# usage: run_variants.sh "<EXTRA flags>" ...   (scratch helper for plan sweeps on the GPU box)
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['input_dtype'], d['value'], d['roofline']['all_kernels_ms_per_step'])"; }
for ex in "$@"; do
  echo "=== EXTRA=$ex"
  make -B -C pulseportraiture_amd/csrc EXTRA="$ex" >/dev/null 2>&1 || { echo build failed; continue; }
  python bench.py --no-cpu-baseline | pj
  python bench.py --no-cpu-baseline --input-dtype f32 | pj
  python bench.py --no-cpu-baseline --workload cfg2-512x1024-phiDM | pj
  python bench.py --no-cpu-baseline | pj
done
