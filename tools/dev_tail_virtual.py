"""Stage check of the one-wave ("virtual waves") solve / post-fit kernels: bitwise equality with the multi-wave
kernels and their standalone duration (= the latency of ONE wave working a subint).  (GPU box)
    python tools/dev_tail_virtual.py"""
import argparse
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from pulseportraiture_amd.engine import Engine

KEYS = ("params", "param_errs", "nu_refs", "cov", "chi2", "red_chi2", "snr", "nfeval", "return_code", "npass",
        "scales", "scale_errs", "channel_snrs")
ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05, truth_guesses=False,
                        measured_noise=False, method="trust-ncg")
eng = Engine(0)
dev = torch.device("cuda", 0)
for wl, nsub in (("toa-4096x2048-phiDM", 1024), ("cfg3-4096x2048-phiDMGM", 1024), ("cfg2-512x1024-phiDM", 1024)):
    b = bench.Batch(eng, ns, dev, wl, nsub, "f64", 0)
    for method in ("trust-ncg", "newton"):
        out = {}
        for tv in (0, 1, 0, 1):
            eng.set_option("tail_virtual", tv)
            eng.set_option("profile", 1)
            eng.kernel_times(reset=True)
            r = eng.fit_batch(b.data, b.freqs, b.P, b.x0, errs=b.errs_dev, nu_fits=np.full((nsub, 3), b.nu_fit),
                              fit_flags=b.flags, per_channel=True, method=method)
            kt = eng.kernel_times(reset=True)
            eng.set_option("profile", 0)
            out.setdefault(tv, r)
            print("%-24s %-9s tail_virtual=%d  solve %.3f ms  finalize %.3f ms  xspec %.3f ms" % (
                wl, method, tv, 1e3 * kt["taylor_solve"][0], 1e3 * kt["finalize"][0], 1e3 * kt["xspec"][0]))
        same = all(np.array_equal(np.asarray(out[0][k]), np.asarray(out[1][k])) for k in KEYS)
        print("   bitwise equal: %s" % same)
        if not same:
            for k in KEYS:
                a, c = np.asarray(out[0][k]), np.asarray(out[1][k])
                if not np.array_equal(a, c):
                    print("     %s differs in %d entries, max |d| %.3e" % (k, int((a != c).sum()), float(np.nanmax(np.abs(a - c)))))
    b.free()
eng.set_option("tail_virtual", 0)
