#!/bin/bash
# usage: tools/pmc_counters.sh <tag> [bench args...] ; collects SQ counter passes for one bench step
# (PP_PMC_KERNELS="k_taylor_solve,k_finalize" picks other kernels than the transform / evaluation)
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/p2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/p2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/p3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/p3.err
python3 - $out <<'PY'
import csv,glob,collections,sys,os
want=os.environ.get("PP_PMC_KERNELS","k_xspec,k_eval").split(",")
for p in ("p1","p2","p3"):
    fs=glob.glob(sys.argv[1]+"/"+p+"/*/*_counter_collection.csv")
    if not fs: print("no output for",p); continue
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if any(w in k for w in want):
            agg[(k[:40],r["Counter_Name"])]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(k, "%.4g"%v)
    ks=glob.glob(sys.argv[1]+"/"+p+"/*/*_kernel_trace.csv")
    if ks:
        for r in csv.DictReader(open(ks[0])):
            if any(w in r["Kernel_Name"] for w in want):
                print("  %s duration ms" % r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
PY
