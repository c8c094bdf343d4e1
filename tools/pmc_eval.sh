#!/bin/bash
# GPU box: counters of the scattering evaluator (k_eval_scat) over three configs[3] fits.
#   tools/pmc_eval.sh <tag>   -> gpurun_out/pmc_eval_<tag>/summary.txt
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_eval_$tag
mkdir -p $out
i=0
for set in \
  "TCC_HIT_sum TCC_MISS_sum" \
  "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" ; do
  i=$((i+1))
  timeout -k 5 ${PP_PMC_TIMEOUT:-60} rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 tools/dev_eval_pass_time.py "$@" > $out/p$i.log 2>&1
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv,glob,collections,sys
for p in sorted(glob.glob(sys.argv[1]+"/p?")):
    fs=glob.glob(p+"/*/*_counter_collection.csv")
    if not fs: print("no output for",p); continue
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        for pat in ("k_eval_scat","k_scat_model(","k_xspec_qs1024"):
            if pat in k:
                agg[(pat,r["Counter_Name"])]+=float(r["Counter_Value"]); n[(pat,r["Counter_Name"])]+=1
    for k,v in sorted(agg.items()): print("%-16s %-36s %.5g per launch (%d launches)"%(k[0],k[1],v/n[k],n[k]))
PY
