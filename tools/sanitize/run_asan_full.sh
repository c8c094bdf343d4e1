#!/bin/bash
# The WHOLE GPU suite (bench children excluded: PP_NO_BENCH_CHILD) against the ASan + UBSan build of the host side.
out=${1:-gpurun_out/r06_asan_full.log}
cd "$(dirname "$0")/../.." || exit 1
RT=/usr/lib/x86_64-linux-gnu/libasan.so.6
export PP_TOAS_LIB=$PWD/tools/sanitize/libpptoas_hip_asan.so PP_NO_BENCH_CHILD=1
export LD_LIBRARY_PATH=$(python -c "import os, importlib.util as u; print(os.path.join(os.path.dirname(u.find_spec('torch').origin), 'lib'))"):$LD_LIBRARY_PATH
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:detect_odr_violation=0:verify_asan_link_order=0:log_path=gpurun_out/r06_asan_full_report
rm -f gpurun_out/r06_asan_full_report*
{
  echo "# ASan + UBSan (host side) under the whole GPU suite: $RT"
  LD_PRELOAD="$RT /usr/lib/x86_64-linux-gnu/libstdc++.so.6" python -m pytest tests/test_gpu_parity.py tests/test_gpu_y_batch_independence.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15
  echo "exit code ${PIPESTATUS[0]}"
  echo "# sanitizer report files: $(ls gpurun_out/r06_asan_full_report* 2>/dev/null | wc -l)"
  cat gpurun_out/r06_asan_full_report* 2>/dev/null | head -100
} > $out 2>&1
tail -12 $out
