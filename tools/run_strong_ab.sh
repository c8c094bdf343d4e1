for rep in 1 2 3; do for o in 1 0; do python bench.py --no-cpu-baseline --no-other-workloads --total-nsub 3000 --opt fuse_tail=$o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fuse_tail=$o', d['value'], d['ms_per_step'], d.get('scaling'), {k:d[k] for k in d if k in ('steps','warmup')}, d['config'].get('sub_batches'))"; done; done
python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('weak', d['value'], d['ms_per_step'])"
