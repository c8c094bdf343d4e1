"""stdin: one bench.py JSON line; prints fits/s and the per-step times of the kernels behind the transform."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["roofline"]["all_kernels_ms_per_step"]
print("%-10s %-24s %9.1f fits/s  ms/step %.4f  taylor_solve %.4f finalize %.4f xspec %.3f" % (
    tag, d["config"]["workload"], d["value"], d["ms_per_step"], k.get("taylor_solve", 0), k.get("finalize", 0), k.get("xspec", 0)))
