#!/bin/bash
# GPU box: shader clock during the transform kernel = GRBM_GUI_ACTIVE / (8 XCDs x duration), for a set of bench arguments
export TMPDIR=/tmp
i=0
for args in "$@"; do
  i=$((i+1)); out=gpurun_out/clk_$i; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads $args > /dev/null 2> $out/err.txt
  python3 - $out "$args" <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
fs=glob.glob(d+"/*/*_counter_collection.csv"); ks=glob.glob(d+"/*/*_kernel_trace.csv")
if not fs or not ks: print("no output", sys.argv[2]); sys.exit()
dur={}
for r in csv.DictReader(open(ks[0])):
    if "k_xspec" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    if "k_xspec" in r["Kernel_Name"]: agg[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"]); agg[r["Dispatch_Id"]]["name"]=r["Kernel_Name"][:40]
for k,v in agg.items():
    ms=dur.get(k,0)
    print("%-44s %-42s %.2f ms  clock %.3f GHz  VALU/row %.0f LDS/row %.0f  LDS busy/wave %.3f" % (sys.argv[2], v["name"], ms, v["GRBM_GUI_ACTIVE"]/8/ms/1e6 if ms else 0, v["SQ_INSTS_VALU"]/4194304, v["SQ_INSTS_LDS"]/4194304, v["SQ_ACTIVE_INST_LDS"]/max(v["SQ_WAVE_CYCLES"],1)))
PY
done
