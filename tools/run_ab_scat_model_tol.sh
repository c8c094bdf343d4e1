out=gpurun_out/r06_scat_model_tol_ab.txt
: > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['convergence']
print('%-34s %9.0f fits/s %8.3f ms/step  %s  nfeval %.2f npass %.2f max %d  checksum %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], c.get('nfeval_mean',0), c.get('npass_mean',0), c.get('npass_max',0), d['gathered_records']['checksum'][:2]))" "$1" >> $out; }
B="python bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 3 --workload cfg4-2048x2048-scat"
for rep in 1 2; do
  for tol in 1e-10 1e-9 1e-8 1e-7 1e-6; do
    $B --opt scat_model_tol=$tol 2>/dev/null | line "scat_model_tol=$tol"
  done
  $B --opt scat_model_bet=0 2>/dev/null | line "scat_model_bet=0"
done
cat $out
