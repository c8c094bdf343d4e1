#!/bin/bash
# timing-only builds of k_xspec (results are wrong by construction, the fit falls back):
# 1 = loads only, 2 = + first FFT stage, 3 = + all stages and S_d, 0 = complete.
# Per-kernel durations from rocprofv3 --kernel-trace --stats (mode-2 instantiation only).
export TMPDIR=/tmp
for a in 1 2 3 0; do
  make -B -C pulseportraiture_amd/csrc EXTRA="-DPP_XSPEC_ABLATE=$a" >/dev/null 2>&1 || { echo build failed; continue; }
  for dt in f64 f32; do
    rm -rf gpurun_out/abl; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -- python3 bench.py --no-cpu-baseline --steps 3 --input-dtype $dt > /dev/null 2>&1
    echo "ABLATE=$a $dt $(grep 'k_xspec' gpurun_out/abl/*/*_kernel_stats.csv | grep ', 2>' | awk -F, '{print $1, "calls", $(NF-6), "avg_ns", $(NF-4)}' | cut -c1-160)"
  done
done
make -B -C pulseportraiture_amd/csrc >/dev/null 2>&1
