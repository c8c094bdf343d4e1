#!/usr/bin/env python3
"""The last unexplained device-only misses of the parity sweep: is the device's answer one the TRUE
reference gives on ANOTHER CPU?  (build container only: imports the true reference from /root/reference.)

tools/ref_exit_points.py refits a case with the true reference under 32 channel orders; for a handful of
fits of few channels no order lands on the device's answer, because what decides their last ratio test
(actual reduction 0 / 1 / 2 ulp(f) against a predicted 1-2 ulp) is the rounding of the sums over HARMONICS,
which a channel permutation does not reorder.  NumPy's own arithmetic changes there with the machine: its
pairwise sums, complex multiplies and exp / sincos are dispatched by SIMD width.  This script refits the
cases in child interpreters started with NPY_DISABLE_CPU_FEATURES (AVX512 off / AVX2 + FMA off as well:
what the reference does on an older host), 32 channel orders each, and reports whether the device's answer
is among the exit points.

    python tools/ref_exit_other_cpu.py [gpurun_out/parity_sweep_rows.json] [case ...] > profiles/r05_ref_exit_other_cpu.txt
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPERM = 32
FEATURE_SETS = [("as built", ""),
                ("AVX512 off", "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX512_KNL AVX512_KNM"),
                ("AVX2 / FMA3 / AVX512 off", "AVX2 FMA3 AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX512_KNL AVX512_KNM")]

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests", "golden"))
from tools.ref_self_scatter import _reference, _fit
from tools.sweep_parity import make_case
ref = _reference()
out = {}
for k in %(cases)r:
    c = make_case(k)
    n = int(c["mask"].sum())
    rng = np.random.default_rng(777 + k)
    orders = [np.arange(n), np.arange(n)[::-1]] + [rng.permutation(n) for _ in range(%(nperm)d - 2)]
    res = []
    for o in orders:
        p, nfev, _ = _fit(ref, c, o)
        res.append([float(p[0]), int(nfev)])
    out[str(k)] = res
feat = getattr(np._core._multiarray_umath, "__cpu_features__", {})
print(json.dumps({"fits": out, "simd": sorted(f for f, on in feat.items() if on)}))
'''


def main():
    jpath = os.path.join(ROOT, "gpurun_out", "parity_sweep_rows.json")
    args = sys.argv[1:]
    if args and args[0].endswith(".json"):
        jpath, args = args[0], args[1:]
    cases = [int(v) for v in args] or [194, 2155]
    dev = {r["k"]: r for r in json.load(open(jpath))["rows"]}
    wrap = lambda d: (d + 0.5) % 1.0 - 0.5
    print("## device-only misses refitted by the TRUE reference with NumPy's SIMD dispatch restricted (%d channel orders each)" % NPERM)
    found = {k: [] for k in cases}
    for label, disable in FEATURE_SETS:
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        if disable:
            env["NPY_DISABLE_CPU_FEATURES"] = disable
        code = CHILD % dict(root=ROOT, cases=cases, nperm=NPERM)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode or not line:
            print("  %-28s FAILED: %s" % (label, p.stderr[-300:]))
            continue
        res = json.loads(line[-1])
        print("NumPy SIMD features in use (%s): %s" % (label, " ".join(res["simd"])))
        for k in cases:
            dphi_dev = dev[k]["params"][0]
            phis = [f[0] for f in res["fits"][str(k)]]
            pts = []
            for ph in phis:
                for q in pts:
                    if abs(wrap(ph - q[0])) < 1e-11:
                        q[1] += 1
                        break
                else:
                    pts.append([ph, 1])
            at_dev = sum(1 for ph in phis if abs(wrap(ph - dphi_dev)) < 1e-11)
            nearest = min(abs(wrap(ph - dphi_dev)) for ph in phis)
            print("  case %5d %s  |dev - ref(natural, this CPU)| %.2e  exit points %d  orders at the device's point %2d of %d  "
                  "nearest reference answer %.1e  nfeval set %s" % (
                      k, dev[k]["flags"], abs(wrap(dphi_dev - phis[0])), len(pts), at_dev, len(phis), nearest,
                      sorted(set(f[1] for f in res["fits"][str(k)]))))
            if at_dev:
                found[k].append(label)
    for k in cases:
        print("case %d: the device's answer is %s" % (
            k, ("an exit point of the true reference with SIMD set: " + ", ".join(found[k])) if found[k]
            else "NOT among the reference's exit points under any of these settings"))


if __name__ == "__main__":
    main()
