#!/usr/bin/env python3
"""Check and tabulate the runs of tools/run_scale.sh:  python tools/check_scale.py OUTDIR N...

Per N and mode: the JSON line parses, n_gpus = N, the gathered rows are what the mode promises, every
return code is 2 (SciPy's status of a converged trust-ncg walk), and
  strong  the gathered [total, 18] records equal the N = 1 job's BIT FOR BIT (shards and sub-batches fall
          differently for every N; a subint's answer does not depend on them)
  weak    rank r fits subints [r nsub, (r + 1) nsub): the checksum of the N-rank job's first nsub * steps rows'
          share is not separable from the line, so the check is rows, return codes and the injected-DM pull.
Exit status 1 if any check fails.  Prints one table (fits/s as bench.py reports them -- no efficiency figure)."""
import json
import os
import sys

import numpy as np

out = sys.argv[1]
ns = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8]
bad = 0
rows = []
base = None
for n in ns:
    for mode in ("weak", "strong"):
        path = os.path.join(out, "%s_n%d.json" % (mode, n))
        if not os.path.exists(path):
            continue
        try:
            line = json.loads(open(path).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            rows.append((mode, n, "NO JSON LINE (see %s_n%d.err)" % (mode, n)))
            bad += 1
            continue
        notes = []
        if line.get("n_gpus") != n or line.get("scaling") != mode:
            notes.append("n_gpus/scaling mismatch")
        g = line.get("gathered_records", {})
        if mode == "weak":
            want = n * line["config"]["nsub_per_gpu_per_step"] * line["steps"]
            if g.get("rows") != want:
                notes.append("rows %s != %d" % (g.get("rows"), want))
            rc = line.get("convergence", {}).get("return_codes", {})
            if set(rc) != {"2"}:
                notes.append("return codes %s" % rc)
        else:
            total = line["config"]["total_nsub"]
            if g.get("rows") != total:
                notes.append("rows %s != %d" % (g.get("rows"), total))
            if g.get("return_code_sum") != 2.0 * total:
                notes.append("return_code_sum %s" % g.get("return_code_sum"))
            if sum(line["config"]["fits_per_rank"]) != total or len(line["config"]["fits_per_rank"]) != n:
                notes.append("shards %s" % line["config"]["fits_per_rank"])
            rp = os.path.join(out, "records_strong_n%d.npy" % n)
            if os.path.exists(rp):
                rec = np.load(rp)
                if base is None:
                    base = rec
                elif rec.shape != base.shape or not np.array_equal(rec, base):
                    diff = int((rec != base).any(axis=1).sum()) if rec.shape == base.shape else -1
                    notes.append("records differ from the N = %d job in %d rows" % (ns[0], diff))
                else:
                    notes.append("records = N=%d job, bit for bit" % ns[0])
        if line.get("max_abs_dDM_over_err", 0) > 6.0:
            notes.append("injected DM off by %.1f sigma" % line["max_abs_dDM_over_err"])
        ok = all("differ" not in x and "!=" not in x and "mismatch" not in x and "codes" not in x and "off by" not in x
                 and "shards" not in x and "sum" not in x for x in notes)
        bad += 0 if ok else 1
        rows.append((mode, n, "%10.0f fits/s  %9.3f ms/step  %s%s" % (
            line["value"], line["ms_per_step"], "OK" if ok else "FAIL", ("  [" + "; ".join(notes) + "]") if notes else "")))
for mode, n, text in rows:
    print("%-6s N=%d  %s" % (mode, n, text))
sys.exit(1 if bad else 0)
