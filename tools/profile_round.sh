#!/bin/bash
# usage (on the MI355X box, repo root):  tools/profile_round.sh r03
# Runs every profiling pass behind profiles/<round>_* and writes the summaries to
# gpurun_out/profiles_<round>/ (copy them into profiles/ afterwards).
# Counter passes are separate runs with --kernel-trace only (no sys/hip traces).
set -u
export TMPDIR=/tmp
R=${1:-r03}
O=gpurun_out/profiles_$R
rm -rf $O; mkdir -p $O/raw
# the driver's command: headline + other workloads + CPU baseline in one line
python3 bench.py > $O/${R}_bench.json 2> $O/raw/bench.err
# profiled passes: headline only (no CPU pool: nothing may be spawned under rocprofv3)
B="python3 bench.py --no-cpu-baseline --no-other-workloads"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/trace -- $B --steps 3 --warmup 1 > $O/${R}_bench_under_rocprof.json 2> $O/raw/trace.err
# (THREE enqueued steps: the timed launches of the transform carry the previous step's solve + post-fit stage as
# tickets -- option fuse_tail -- so the figure that belongs beside the timed kernel is the SECOND dispatch's, not the
# first's, which carries nothing; both are written to traffic_latest.json)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_fetch -- $B --steps 3 --warmup 0 > /dev/null 2> $O/raw/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_write -- $B --steps 3 --warmup 0 > /dev/null 2> $O/raw/write.err
tools/pmc_counters.sh $R --no-other-workloads > $O/${R}_sq_counters.txt 2>&1
# configs[3] (scattering): kernel stats and HBM counters of its own
W4="--workload cfg4-2048x2048-scat"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/trace_cfg4 -- $B $W4 --steps 3 --warmup 1 > $O/raw/bench_cfg4_under_rocprof.json 2> $O/raw/trace4.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_fetch_cfg4 -- $B $W4 --steps 1 --warmup 0 > /dev/null 2> $O/raw/fetch4.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_write_cfg4 -- $B $W4 --steps 1 --warmup 0 > /dev/null 2> $O/raw/write4.err
# configs[1] (512 x 1024, k_xspec_qf<512>) and the masked regime (20 % of the rows skipped): kernel stats + HBM counters
for tag in cfg2 masked20; do
  if [ $tag = cfg2 ]; then WX="--workload cfg2-512x1024-phiDM"; else WX="--variant masked20"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/trace_$tag -- $B $WX --steps 5 --warmup 1 > $O/raw/bench_${tag}_under_rocprof.json 2> $O/raw/trace_$tag.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_fetch_$tag -- $B $WX --steps 1 --warmup 0 > /dev/null 2> $O/raw/fetch_$tag.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_write_$tag -- $B $WX --steps 1 --warmup 0 > /dev/null 2> $O/raw/write_$tag.err
done
# the seeded (get_TOAs) flows: the device seed, and the reference's own guess formed inside the single pass
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/trace_seeded -- $B --seed-ns 100 --steps 3 --warmup 1 > $O/raw/bench_seeded_under_rocprof.json 2> $O/raw/trace_s.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/trace_refseed -- $B --seed-ns -1 --steps 3 --warmup 1 > $O/raw/bench_refseed_under_rocprof.json 2> $O/raw/trace_r.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_fetch_refseed -- $B --seed-ns -1 --steps 1 --warmup 0 > /dev/null 2> $O/raw/fetch_r.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/raw/pmc_write_refseed -- $B --seed-ns -1 --steps 1 --warmup 0 > /dev/null 2> $O/raw/write_r.err
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
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
    for r in rows:
        e = out.setdefault(r["Kernel_Name"], {"dispatches": 0, "sum_KiB": 0.0, "per_dispatch_KiB": []})
        e["dispatches"] += 1
        e["sum_KiB"] += float(r["Counter_Value"])
        e["per_dispatch_KiB"].append(float(r["Counter_Value"]))
    return out
bench = json.loads(open(f"{O}/{R}_bench.json").read().strip().splitlines()[-1])
nsub = bench["config"]["nsub_per_gpu_per_step"]
fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
json.dump({"units": f"KiB, summed and per dispatch, of three enqueued bench steps ({nsub} fits of {bench['config']['nchan']}x{bench['config']['nbin']} {bench['config']['input_dtype']})",
           "note": "gfx950 FETCH_SIZE counts half of a wide coalesced read: double it (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact",
           "fetch": fetch, "write": write}, open(f"{O}/{R}_pmc_hbm_counters.json", "w"), indent=1)
fam = bench["roofline"]["kernel"]
kname = {"xspec": "k_xspec", "eval": "k_eval"}.get(fam, fam)
# the dominant transform kernel's dispatches in order: [0] carries no tail, [1] and [2] carry the previous step's
def per_dispatch(tab, scale):
    best = max(((k, v) for k, v in tab.items() if kname in k), key=lambda kv: kv[1]["sum_KiB"], default=(None, None))[1]
    return [x * scale for x in best["per_dispatch_KiB"]] if best else []
fd, wd = per_dispatch(fetch, 2048.0), per_dispatch(write, 1024.0)
fb = sum(fd)
if fb > 0 and len(fd) == len(wd):
    tot = [a + b for a, b in zip(fd, wd)]
    with_tail = tot[1] if len(tot) > 1 else tot[0]
    # (the stand-alone solve + post-fit kernels of the LAST step, which nobody carries, for comparison)
    alone = sum(v["per_dispatch_KiB"][-1] * 2048.0 for k, v in fetch.items() if "k_taylor_solve" in k or "k_finalize" in k) + \
            sum(v["per_dispatch_KiB"][-1] * 1024.0 for k, v in write.items() if "k_taylor_solve" in k or "k_finalize" in k)
    tl = {"workload": bench["config"]["workload"], "input_dtype": bench["config"]["input_dtype"], "nsub": nsub,
          "kernel": fam, "hbm_bytes_per_launch": with_tail, "hbm_bytes_per_fit": with_tail / nsub,
          "hbm_bytes_per_launch_without_tail": tot[0], "hbm_bytes_per_fit_without_tail": tot[0] / nsub,
          "dispatches_profiled": len(tot), "hbm_bytes_per_launch_each": tot,
          "stand_alone_solve_and_post_fit_bytes": alone,
          "note": "hbm_bytes_per_launch is the SECOND of three enqueued steps' transform: like every timed launch it carries the "
                  "previous step's solve + post-fit stage as tickets (fuse_tail); the first dispatch carries none",
          "source": f"profiles/{R}_pmc_hbm_counters.json (FETCH_SIZE x2 + WRITE_SIZE, three enqueued steps)"}
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
    try:
        # GRBM_GUI_ACTIVE / 8 XCDs / duration of the same counter pass
        g = cval("GRBM_GUI_ACTIVE")
        durs = [float(x) for x in re.findall(r"xspec[^\n]* duration ms ([0-9.]+)", txt)]
        if durs:
            tl["shader_clock_ghz"] = round(g / 8 / (durs[-1] * 1e6), 3)
            tl["counters_source"] += "; GRBM_GUI_ACTIVE / 8 / duration"
    except Exception as ex:
        print("no clock:", ex)
    json.dump(tl, open(f"{O}/traffic_latest.json", "w"), indent=1)
# configs[3]: per-launch traffic of the evaluator and of the transform
def first(pat):
    fs = glob.glob(pat)
    return fs[0] if fs else None
for tag in ("cfg4", "seeded", "refseed", "cfg2", "masked20"):
    f = first(f"{O}/raw/trace_{tag}/*/*_kernel_stats.csv")
    if f:
        open(f"{O}/{R}_{tag}_kernel_stats.csv", "w").write(open(f).read())
# configs[1] and the masked regime: HBM traffic of the transform against the algorithmic bytes
for tag in ("cfg2", "masked20"):
    fx, wx = counters(f"pmc_fetch_{tag}", "FETCH_SIZE"), counters(f"pmc_write_{tag}", "WRITE_SIZE")
    if not fx:
        continue
    try:
        bx = json.loads(open(f"{O}/raw/bench_{tag}_under_rocprof.json").read().strip().splitlines()[-1])
        nx = bx["config"]["nsub_per_gpu_per_step"]
        fb = sum(v["sum_KiB"] for k, v in fx.items() if "k_xspec" in k) * 2048
        wb = sum(v["sum_KiB"] for k, v in wx.items() if "k_xspec" in k) * 1024
        kn = [k for k in fx if "k_xspec" in k]
        json.dump({"workload": bx["config"]["workload"], "variant": bx["config"].get("variant"), "nsub": nx,
                   "transform_kernels": kn, "fits_per_s_under_profiler": bx["value"],
                   "algorithmic_bytes_per_fit": bx["roofline"]["algorithmic_bytes_per_fit"],
                   "transform_hbm_bytes_per_fit": (fb + wb) / nx,
                   "traffic_over_algorithmic": (fb + wb) / nx / bx["roofline"]["algorithmic_bytes_per_fit"],
                   "roofline_under_profiler": {k: bx["roofline"][k] for k in ("achieved", "frac", "ms_per_step_in_kernel")},
                   "note": "FETCH_SIZE x2 + WRITE_SIZE of the transform kernel of one step (separate --pmc passes)"},
                  open(f"{O}/{R}_{tag}_traffic.json", "w"), indent=1)
    except Exception as ex:
        print(tag, "traffic summary failed:", ex)
f4, w4 = counters("pmc_fetch_cfg4", "FETCH_SIZE"), counters("pmc_write_cfg4", "WRITE_SIZE")
if f4:
    try:
        b4 = json.loads(open(f"{O}/raw/bench_cfg4_under_rocprof.json").read().strip().splitlines()[-1])
        n4 = b4["config"]["nsub_per_gpu_per_step"]
        alg = b4["roofline"]["algorithmic_bytes_per_fit"]
        rows = {}
        for kn in ("k_eval", "k_xspec", "k_accum", "k_scat_model(", "k_scat_model_solve", "k_step", "k_finalize"):
            fb = sum(v["sum_KiB"] for k, v in f4.items() if kn in k) * 2048
            wb = sum(v["sum_KiB"] for k, v in w4.items() if kn in k) * 1024
            nd = sum(v["dispatches"] for k, v in f4.items() if kn in k)
            rows[kn] = {"dispatches_per_step": nd, "hbm_bytes_per_fit": (fb + wb) / n4,
                        "hbm_bytes_per_fit_per_dispatch": (fb + wb) / n4 / max(nd, 1)}
        tot = sum(r["hbm_bytes_per_fit"] for r in rows.values())
        json.dump({"workload": b4["config"]["workload"], "method": b4["config"]["method"], "nsub": n4,
                   "fits_per_s_under_profiler": b4["value"], "algorithmic_bytes_per_fit": alg,
                   "hbm_bytes_per_fit_all_kernels": tot, "traffic_over_algorithmic": tot / alg,
                   "kernels": rows, "note": "FETCH_SIZE x2 + WRITE_SIZE of one step (separate --pmc passes)"},
                  open(f"{O}/{R}_cfg4_traffic.json", "w"), indent=1)
    except Exception as ex:
        print("cfg4 traffic summary failed:", ex)
# the reference-seed flow: traffic of its one pass (and of everything else in the step)
fr, wr = counters("pmc_fetch_refseed", "FETCH_SIZE"), counters("pmc_write_refseed", "WRITE_SIZE")
if fr:
    try:
        br = json.loads(open(f"{O}/raw/bench_refseed_under_rocprof.json").read().strip().splitlines()[-1])
        nr = br["config"]["nsub_per_gpu_per_step"]
        rows = {}
        for k in sorted(set(fr) | set(wr)):
            # (not the untimed batch generation: k_synth, the template, the generation's own guesses)
            if any(x in k for x in ("k_synth", "k_model_", "k_rot_mean", "vectorized_elementwise")):
                continue
            fb = fr.get(k, {"sum_KiB": 0.0})["sum_KiB"] * 2048
            wb = wr.get(k, {"sum_KiB": 0.0})["sum_KiB"] * 1024
            nd = fr.get(k, wr.get(k))["dispatches"]
            if ("k_fps" in k or "k_rfft_rows" in k) and nd == 2:    # (one of the two belongs to the generation)
                fb, wb, nd = fb / 2, wb / 2, 1
            if fb + wb > 1e6 * nr * 0.01:
                rows[k[:60]] = {"hbm_bytes_per_fit": (fb + wb) / nr, "dispatches": nd}
        tot = sum(r["hbm_bytes_per_fit"] for r in rows.values())
        alg = br["roofline"]["algorithmic_bytes_per_fit"]
        json.dump({"workload": br["config"]["workload"], "phase_guesses": br["config"]["phase_guesses"], "nsub": nr,
                   "fits_per_s_under_profiler": br["value"], "algorithmic_bytes_per_fit": alg,
                   "hbm_bytes_per_fit_all_kernels": tot, "traffic_over_algorithmic": tot / alg, "kernels": rows,
                   "note": "FETCH_SIZE x2 + WRITE_SIZE of one step (separate --pmc passes)"},
                  open(f"{O}/{R}_refseed_traffic.json", "w"), indent=1)
    except Exception as ex:
        print("refseed traffic summary failed:", ex)
PY
ls -la $O
# shader clock under the transform kernels (power cap) and the one-exchange kernel against the general one
./tools/run_clock_probe.sh "--opt one_exchange=0" "--opt one_exchange=1" "--input-dtype f32 --opt one_exchange=1" "--seed-ns -1" "--seed-ns -1 --input-dtype f32" > $O/${R}_clock_probe.txt 2>&1
./tools/run_ab_option.sh one_exchange > $O/${R}_one_exchange_ab.txt 2>&1
