"""Host time of each pp_fit_enqueue and wait of each pp_fit_collect over 40 pipelined steps of configs[1]
(argv: profile 0/1 [records]).  (GPU box)"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from pulseportraiture_amd.engine import Engine
ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05, truth_guesses=False,
                        measured_noise=False, method="trust-ncg")
eng = Engine(0); dev = torch.device("cuda", 0)
b = bench.Batch(eng, ns, dev, "cfg2-512x1024-phiDM", 1024, "f64", 0)
for _ in range(3): b.fit()
eng.synchronize(); torch.cuda.synchronize()
prof = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng.set_option("profile", prof)
N = 40
from pulseportraiture_amd import dist as ppdist
use_recs = len(sys.argv) > 2
recs = torch.zeros((N, 1024, ppdist.RECORD_WIDTH), dtype=torch.float64, device=dev)
te, tc = [], []
t00 = time.perf_counter()
for k in range(N):
    t0 = time.perf_counter(); b.enqueue(records=recs[k] if use_recs else None); te.append(time.perf_counter() - t0)
    if k > 0:
        t0 = time.perf_counter(); eng.collect(); tc.append(time.perf_counter() - t0)
t0 = time.perf_counter(); eng.collect(); tc.append(time.perf_counter() - t0)
tot = time.perf_counter() - t00
print("profile", prof, "total %.2f ms, per step %.3f" % (1e3 * tot, 1e3 * tot / N))
print("enqueue ms:", " ".join("%.2f" % (1e3 * x) for x in te))
print("collect ms:", " ".join("%.2f" % (1e3 * x) for x in tc))
