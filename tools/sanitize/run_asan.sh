#!/bin/bash
# On the GPU box: the GPU suite's host-heavy tests against the AddressSanitizer + UBSan build of the library's HOST
# side (make -C tools/sanitize asan, built beforehand: it travels with the snapshot).  Device code is the product's;
# no GPU sanitizer, no XNACK.   bash tools/sanitize/run_asan.sh [out]
out=${1:-gpurun_out/r06_asan.log}
cd "$(dirname "$0")/../.." || exit 1
RT=$(ls /usr/lib/x86_64-linux-gnu/libasan.so.6 | head -1)     # (GCC's runtime: see the Makefile)
export PP_TOAS_LIB=$PWD/tools/sanitize/libpptoas_hip_asan.so
# (ASan's dlopen interceptor makes torch's lazy dlopen of its own libraries miss their RUNPATH: name the directory)
export LD_LIBRARY_PATH=$(python -c "import os, importlib.util as u; print(os.path.join(os.path.dirname(u.find_spec('torch').origin), 'lib'))"):$LD_LIBRARY_PATH
# (libstdc++ preloaded with the runtime: its __cxa_throw interceptor must find the real one, and the interpreter does not
# link the C++ runtime; detect_leaks=0: the interpreter and the HIP runtime keep their allocations; protect_shadow_gap=0: the HSA runtime maps
# device-visible memory where ASan expects its shadow gap)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:detect_odr_violation=0:verify_asan_link_order=0:log_path=gpurun_out/r06_asan_report
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
{
  echo "# ASan + UBSan (host side) under pytest on the GPU box: $RT"
  LD_PRELOAD="$RT /usr/lib/x86_64-linux-gnu/libstdc++.so.6" python -m pytest tests/test_gpu_parity.py tests/test_gpu_y_batch_independence.py -q -m gpu -p no:cacheprovider \
      -k "enqueue or submit or tail or sub_batching or degenerate or options_match or ragged or masked_rows" 2>&1 | tail -40
  echo "exit code ${PIPESTATUS[0]}"
  echo "# sanitizer report files: $(ls gpurun_out/r06_asan_report* 2>/dev/null | wc -l)"
  cat gpurun_out/r06_asan_report* 2>/dev/null | head -200
} > $out 2>&1
grep -c "ERROR: AddressSanitizer\|runtime error:" $out
tail -15 $out
