# per-launch time of the scattering evaluator on configs[3] (2048 x 2048, 512 subints)
import sys, numpy as np
sys.path.insert(0, '.')
import argparse, bench, torch
from pulseportraiture_amd.engine import Engine
eng = Engine(0)
args = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05, truth_guesses=False,
                          measured_noise=False, method=sys.argv[1] if len(sys.argv) > 1 else "newton", two_pass_seed=False)
b = bench.Batch(eng, args, torch.device("cuda:0"), "cfg4-2048x2048-scat", 0, "f64", 0)
b.fit()
eng.set_option("profile", 1); eng.kernel_times(reset=True)
for _ in range(3):
    b.fit()
kt = eng.kernel_times(reset=True)
for k, (ms, n) in sorted(kt.items()):
    print("%-14s %8.3f ms in %3d launches = %.4f ms each" % (k, 1e3 * ms, n, 1e3 * ms / max(n, 1)))
