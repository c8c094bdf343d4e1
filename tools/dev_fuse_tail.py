"""Stage check of option fuse_tail (the solve + post-fit stage of an enqueued batch worked off as tickets by the NEXT
batch's transform): enqueued three deep, every batch must return, bit for bit, what a synchronous call returns -- plain
fits, masks + measured noise, GM, Newton, a batch with poor guesses (collect re-fits it), a flow that cannot carry a
tail (scattering) in between, and a last batch nobody follows (flush).  (GPU box)   python tools/dev_fuse_tail.py"""
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
bad = 0
for wl, nsub in (("toa-4096x2048-phiDM", 300), ("cfg2-512x1024-phiDM", 700), ("cfg3-4096x2048-phiDMGM", 130)):
    b = bench.Batch(eng, ns, dev, wl, nsub, "f64", 0)
    rng = np.random.default_rng(3)
    mask = torch.from_numpy((rng.random((nsub, b.C)) > 0.2).astype(np.uint8)).to(dev)
    x_poor = b.x0.copy()
    x_poor[[3, 17], 0] = (x_poor[[3, 17], 0] + 0.03 + 0.5) % 1 - 0.5
    base = dict(errs=b.errs_dev, nu_fits=np.full((nsub, 3), b.nu_fit), fit_flags=b.flags, per_channel=True)
    jobs = [(b.x0, dict(base)), (b.x0, dict(base, chan_mask=mask, errs=None)), (x_poor, dict(base)),
            (b.x0, dict(base, method="newton")), (b.x0[:nsub // 2], None), (b.x0, dict(base)), (b.x0, dict(base, chan_mask=mask))]

    def call(fn, x, kw):
        if kw is None:       # half a batch: another size in between
            k2 = dict(base, errs=b.errs_dev[:nsub // 2], nu_fits=np.full((nsub // 2, 3), b.nu_fit))
            return fn(b.data[:nsub // 2], b.freqs, b.P[:nsub // 2], x, **k2)
        return fn(b.data, b.freqs, b.P, x, **kw)
    eng.set_option("fuse_tail", 0)
    sync = [call(eng.fit_batch, x, kw) for x, kw in jobs]
    for ft, depth in ((1, 3), (1, 2), (0, 3)):
        eng.set_option("fuse_tail", ft)
        got = []
        for j, (x, kw) in enumerate(jobs):
            call(eng.enqueue, x, kw)
            if j >= depth - 1:
                got.append(eng.collect())
        while len(got) < len(jobs):
            got.append(eng.collect())
        for j, (a, g) in enumerate(zip(sync, got)):
            same = all(np.array_equal(np.asarray(a[k]), np.asarray(g[k])) for k in KEYS)
            if not same:
                bad += 1
                print("%s fuse_tail=%d depth %d job %d: NOT equal" % (wl, ft, depth, j))
                for k in KEYS:
                    p, q = np.asarray(a[k]), np.asarray(g[k])
                    if not np.array_equal(p, q):
                        print("    %s differs in %d entries (max |d| %.3e)" % (k, int((p != q).sum()), float(np.nanmax(np.abs(p.astype(float) - q.astype(float))))))
        print("%s fuse_tail=%d depth %d: %d jobs compared" % (wl, ft, depth, len(jobs)))
    b.free()
eng.set_option("fuse_tail", 0)
print("MISMATCHES: %d" % bad)
