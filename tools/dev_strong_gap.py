"""Why a sub-batch of the strong-scaling flow (generate, sync, fit synchronously) costs more than a step of the
weak loop: time a fit right after the generator and the same fit again at once.  (GPU box)"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from pulseportraiture_amd.engine import Engine

ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05, truth_guesses=False,
                        measured_noise=False, method="trust-ncg")
eng = Engine(0)
dev = torch.device("cuda", 0)
b = bench.Batch(eng, ns, dev, "toa-4096x2048-phiDM", 1024, "f64", 0)
def sync():
    eng.synchronize(); torch.cuda.synchronize()
b.fit(); sync()
for rep in range(3):
    b.generate(1024 * (rep + 1)); sync()
    ts = []
    for k in range(3):
        t0 = time.perf_counter(); r = b.fit(); sync(); ts.append(1e3 * (time.perf_counter() - t0))
    time.sleep(0.05); sync()
    t0 = time.perf_counter(); r = b.fit(); sync(); t_idle = 1e3 * (time.perf_counter() - t0)
    print("after generate: %.2f ms, again: %.2f, %.2f; after 50 ms idle: %.2f; engine duration %.2f" % (ts[0], ts[1], ts[2], t_idle, 1e3 * float(r["duration"])))
