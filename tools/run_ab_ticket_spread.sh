out=gpurun_out/r06_ticket_spread_ab.txt
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-34s %9.0f fits/s %8.3f ms/step  xspec %.3f  checksum %s' % (sys.argv[1], sys.argv[2] or 'headline', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['xspec'], d['gathered_records']['checksum'][:2]))" "$1" "$2" >> $out; }
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 30 --warmup 3 --pipeline 3"
for rep in 1 2; do
  for wl in "" "--workload cfg3-4096x2048-phiDMGM" "--variant masked20" "--input-dtype f32"; do
    $B $wl 2>/dev/null | line "front-loaded" "$wl"
    PP_TOAS_LIB=variants/spread.so $B $wl 2>/dev/null | line "spread" "$wl"
  done
done
cat $out
