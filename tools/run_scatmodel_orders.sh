#!/bin/bash
# configs[3] with the scattering model at several degrees (rebuilds the library; GPU box)
export TMPDIR=/tmp
for P in "$@"; do
  make -B -C pulseportraiture_amd/csrc EXTRA="-DPP_MP=$P" >/dev/null 2>&1 || { echo build failed; continue; }
  echo "=== PP_MP=$P"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sm_trace_$P -- python3 bench.py --no-cpu-baseline --no-other-workloads --workload cfg4-2048x2048-scat --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['convergence'])"
  f=$(ls -t gpurun_out/sm_trace_$P/*/*_kernel_stats.csv | head -1); cut -d, -f1-7 $f | grep "scat_model\|k_eval\|k_step"
  rm -rf gpurun_out/sm_trace_$P
done
make -B -C pulseportraiture_amd/csrc >/dev/null 2>&1
