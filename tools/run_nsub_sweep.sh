out=gpurun_out/r06_nsub_sweep.txt
: > $out
for rep in 1 2; do
for ns in 896 960 1024 1032 1040 1056 1088 1152 1280; do
  python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 3 --nsub $ns 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ns=d['config']['nsub_per_gpu_per_step']; k=d['roofline']['ms_per_step_in_kernel']
print('nsub %5d  %8.0f fits/s  step %.3f ms  xspec %.3f ms  -> %.4f us per fit in the kernel, chunks per wave %.3f' % (ns, d['value'], d['ms_per_step'], k, 1e3*k/ns, ns*4096/32/4096.0))" >> $out
done
done
cat $out
