#!/bin/bash
# usage (on the MI355X box, repo root):  ./tools_profile.sh r01
# Runs every profiling pass behind profiles/<round>_* and writes the summaries to
# gpurun_out/profiles_<round>/ (copy them into profiles/ afterwards).
# Counter passes are separate runs with --kernel-trace only (no sys/hip traces).
set -u
export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/profiles_$R
rm -rf $O; mkdir -p $O/raw
python3 bench.py > $O/${R}_bench.json 2> $O/raw/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_bench_under_rocprof.json 2> $O/raw/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/raw/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/raw/write.err
./tools_pmc.sh $R > $O/${R}_sq_counters.txt 2>&1
for w in cfg2-512x1024-phiDM cfg3-4096x2048-phiDMGM cfg4-2048x2048-scat; do
  python3 bench.py --workload $w > $O/raw/bench_$w.json 2>> $O/raw/bench.err
done
python3 bench.py --input-dtype f32 --no-cpu-baseline > $O/raw/bench_f32.json 2>> $O/raw/bench.err
python3 bench.py --seed-ns 100 --no-cpu-baseline > $O/raw/bench_seeded.json 2>> $O/raw/bench.err
python3 - $O $R <<'PY'
import csv, glob, json, sys, collections, os
O, R = sys.argv[1], sys.argv[2]
# kernel stats
ks = glob.glob(O + "/raw/trace/*/*_kernel_stats.csv")
if ks:
    open(f"{O}/{R}_kernel_stats.csv", "w").write(open(ks[0]).read())
def counters(d, name):
    out = collections.OrderedDict()
    fs = glob.glob(f"{O}/raw/{d}/*/*_counter_collection.csv")
    if not fs:
        return out
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != name:
            continue
        e = out.setdefault(r["Kernel_Name"], {"dispatches": 0, "sum_KiB": 0.0})
        e["dispatches"] += 1
        e["sum_KiB"] += float(r["Counter_Value"])
    return out
bench = json.loads(open(f"{O}/{R}_bench.json").read().strip().splitlines()[-1])
nsub = bench["config"]["nsub_per_gpu_per_step"]
fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
json.dump({"units": f"KiB per dispatch group of one bench step ({nsub} fits of {bench['config']['nchan']}x{bench['config']['nbin']} {bench['config']['input_dtype']})",
           "note": "gfx950 FETCH_SIZE counts half of a wide coalesced read: double it (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact",
           "fetch": fetch, "write": write}, open(f"{O}/{R}_pmc_hbm_counters.json", "w"), indent=1)
fam = bench["roofline"]["kernel"]
kname = {"xspec": "k_xspec", "eval": "k_eval"}.get(fam, fam)
fb = sum(v["sum_KiB"] for k, v in fetch.items() if kname in k) * 1024 * 2
wb = sum(v["sum_KiB"] for k, v in write.items() if kname in k) * 1024
nl = max(1, max([v["dispatches"] for k, v in fetch.items() if kname in k] or [1]))
if fb > 0:
    tl = {"workload": bench["config"]["workload"], "input_dtype": bench["config"]["input_dtype"], "nsub": nsub,
          "kernel": fam, "hbm_bytes_per_launch": (fb + wb) / nl, "hbm_bytes_per_fit": (fb + wb) / nl / nsub,
          "source": f"profiles/{R}_pmc_hbm_counters.json (FETCH_SIZE x2 + WRITE_SIZE)"}
    try:
        import re
        txt = open(f"{O}/{R}_sq_counters.txt").read()
        def cval(name):
            return float(re.search(r"%s[^\n]*'%s'\) ([0-9.e+]+)" % (kname, name), txt).group(1))
        tl.update(valu_issue_frac_per_wave=round(cval("SQ_ACTIVE_INST_VALU") / cval("SQ_WAVE_CYCLES"), 4),
                  waves_per_simd=2, valu_insts_per_launch=cval("SQ_INSTS_VALU"),
                  counters_source=f"profiles/{R}_sq_counters.txt (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)")
    except Exception as ex:
        print("no SQ counters for the co-limit:", ex)
    json.dump(tl, open(f"{O}/traffic_latest.json", "w"), indent=1)
allw = {}
for tag, fn in [("plain", f"{O}/{R}_bench.json")] + [(os.path.basename(f)[6:-5], f) for f in sorted(glob.glob(O + "/raw/bench_*.json"))]:
    try:
        d = json.loads(open(fn).read().strip().splitlines()[-1])
    except Exception as ex:
        allw[tag] = {"error": str(ex)}; continue
    allw[tag] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "nsub": d["config"]["nsub_per_gpu_per_step"],
                 "kernels_ms": d["roofline"]["all_kernels_ms_per_step"], "frac": d["roofline"]["frac"],
                 "conv": d.get("convergence"), "cpu": d.get("cpu_baseline")}
json.dump(allw, open(f"{O}/{R}_all_workloads.json", "w"), indent=1)
print(json.dumps({k: (v.get("value"), v.get("kernels_ms")) for k, v in allw.items()}, indent=1))
PY
ls -la $O
