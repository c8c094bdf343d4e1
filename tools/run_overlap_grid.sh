out=gpurun_out/r05_overlap_grid.txt
: > $out
for rep in 1 2; do
 for wl in cfg2-512x1024-phiDM toa-4096x2048-phiDM; do
  for cfg in "1.0 0" "0.9375 1" "0.875 1" "0.96875 1" "1.0 1"; do
    set -- $cfg
    PP_GRID_SCALE=$1 python bench.py --no-cpu-baseline --no-other-workloads --workload $wl --steps 30 --warmup 3 --opt overlap_post=$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s grid x %-8s overlap_post=%s %9.0f fits/s %8.3f ms/step  %s' % ('$wl', '$1', '$2', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']))" >> $out
  done
 done
done
cat $out
