"""Randomised check of the retraced simplex finish: fit_phase_shift on the device
(finish='simplex') against SciPy's own brute + fmin (the oracle) for many random
profiles, bin counts, noise levels, grids.  (GPU box)   python tools/sweep_fps.py [n]"""
import multiprocessing as mp
import os, sys, time
import numpy as np
sys.path.insert(0, ".")


def make(k):
    rng = np.random.default_rng(5000 + k)
    B = int(2 ** rng.integers(6, 13))
    ph = (np.arange(B) + 0.5) / B
    ncomp = int(rng.integers(1, 4))
    prof = np.zeros(B)
    for _ in range(ncomp):
        loc, wid, amp = rng.uniform(0.2, 0.8), rng.uniform(0.005, 0.08), rng.uniform(0.3, 1.0)
        d = (ph - loc + 0.5) % 1.0 - 0.5
        prof += amp * np.exp(-0.5 * (d / wid) ** 2)
    shift = rng.uniform(-0.5, 0.5)
    sigma = float(rng.choice([1e-4, 1e-2, 0.1, 0.5]))
    from oracle import pptoas_oracle as orc
    data = orc.rotate_data(prof, -shift) * rng.uniform(0.5, 3.0) + rng.normal(0, sigma, B)
    noise = None if rng.random() < 0.5 else sigma
    Ns = int(rng.choice([100, 100, 100, 37, 256]))
    return data, prof, noise, Ns


def oracle(k):
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    from oracle import pptoas_oracle as orc
    data, prof, noise, Ns = make(k)
    r = orc.fit_phase_shift(data, prof, noise=noise, Ns=Ns)
    return k, r.phase, r.phase_err, r.scale, r.snr, r.red_chi2


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    t0 = time.time()
    with mp.get_context("spawn").Pool(48) as pool:
        ora = {r[0]: r[1:] for r in pool.imap_unordered(oracle, range(n), chunksize=8)}
    print("oracle: %d fits in %.1f s" % (n, time.time() - t0))
    from pulseportraiture_amd.pplib import fit_phase_shift
    dph, derr, bad = [], [], []
    for k in range(n):
        data, prof, noise, Ns = make(k)
        r = fit_phase_shift(data, prof, noise=noise, Ns=Ns)
        o = ora[k]
        d = abs(r.phase - o[0]); d = min(d, abs(d - 1.0))
        dph.append(d)
        derr.append(max(abs(r.phase_err / o[1] - 1), abs(r.scale / o[2] - 1), abs(r.snr / o[3] - 1), abs(r.red_chi2 / o[4] - 1)))
        if d > 1e-11:
            bad.append((k, len(data), noise, Ns, d))
    dph, derr = np.array(dph), np.array(derr)
    print("simplex finish vs SciPy: |dphase| median %.1e  99%% %.1e  max %.1e ; within 1e-11: %.2f %%" % (
        np.median(dph), np.percentile(dph, 99), dph.max(), 100 * (dph <= 1e-11).mean()))
    print("other fields, max relative difference: median %.1e  99%% %.1e  max %.1e" % (
        np.median(derr), np.percentile(derr, 99), derr.max()))
    for b in bad[:10]:
        print("   outlier: case %d nbin %d noise %s Ns %d |dphase| %.2e" % b)
