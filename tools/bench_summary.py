#!/usr/bin/env python3
"""Print the essentials of one or more bench.py JSON lines:  python tools/bench_summary.py FILE..."""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print("== %s: %.0f fits/s, %.3f ms/step, roofline %.4f (%s %.3f ms), kernels %s" % (
        path, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"],
        d["roofline"]["ms_per_step_in_kernel"], d["roofline"]["all_kernels_ms_per_step"]))
    for k, v in d.get("other_workloads", {}).items():
        if "error" in v:
            print("   %-28s ERROR %s" % (k, v["error"]))
        elif "kernels_ms_per_step" in v:
            ks = v["kernels_ms_per_step"]
            print("   %-28s %9.0f fits/s %8.3f ms  (outside kernels %.3f)  %s" % (
                k, v["fits_per_s"], v["ms_per_step"], v["ms_per_step"] - sum(ks.values()), ks))
        else:
            print("   %-28s %s" % (k, v))
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("   cpu: pool %.2f fits/s (%d workers), one core %.3f; parity %s" % (
            c["value"], c["workers"], c["one_core"]["value"], c["parity_on_sample"]))
