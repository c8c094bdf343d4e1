"""Batch independence: a subint's answer must not depend on what else is in the batch.
Random batches (good, poor and hopeless guesses mixed; random masks; every flag family;
both solvers; with and without the device seed) are fitted whole and subint by subint;
the two must agree to rounding.  Exercises the lists, the compaction, the re-expansion,
the fallbacks and the scattering model together.  (GPU box)   python tools/sweep_batch.py [nbatch]"""
import sys
import numpy as np
import os
LOG2NBIN = tuple(int(v) for v in os.environ.get("PP_SWEEP_LOG2NBIN", "7,11").split(","))   # e.g. "11,12": 2048 only
sys.path.insert(0, ".")
from tests.synth_host import make_inputs, caller_guess, model_portrait
from pulseportraiture_amd.engine import Engine

FLAGS = [([1, 1, 0, 0, 0], False), ([1, 0, 0, 0, 0], False), ([1, 1, 1, 0, 0], False),
         ([1, 1, 0, 1, 1], True), ([1, 1, 0, 1, 0], True), ([1, 0, 0, 1, 1], True)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
eng = Engine(0)
rng = np.random.default_rng(4711)
worst = {}
nfit = 0
for b in range(nb):
    flags, scat = FLAGS[b % len(FLAGS)]
    # (bands up to 512 channels: at 128 channels and more the channel sums are formed in several chunks)
    C = int(rng.integers(8, 40)) if rng.random() < 0.5 else int(rng.integers(64, 513)); nbin = int(2 ** rng.integers(*LOG2NBIN)); N = int(rng.integers(5, 14))
    l10 = bool(rng.random() < 0.6) if scat else False
    freqs, model = model_portrait(C, nbin)
    eng.set_model(model)
    data, x0, errs, masks, nuf, Ps = [], [], [], [], [], []
    for i in range(N):
        tau_us = float(rng.uniform(15, 40)) if scat else None
        inp = make_inputs(C, nbin, 31000 + 100 * b + i, model=model, DM0=(34.56789 if rng.random() < 0.3 else 0.0),
                          sigma=float(rng.choice([0.03, 0.1])), GM=(0.25 if flags[2] else None), tau_us=tau_us)
        g = caller_guess(inp, fit_scat=scat, log10_tau=l10,
                         tau_guess_rot=(1.3 * tau_us * 1e-6 / inp["P"]) if scat else None)
        x = g["init_params"].copy()
        u = rng.random()
        if u < 0.3:                       # poor DM guess: leaves the Taylor range, one re-expansion away
            x[1] += rng.choice([-1, 1]) * rng.uniform(2e-3, 8e-3)
        elif u < 0.45:                    # hopeless phase guess: evaluation fallback
            x[0] = (x[0] + rng.choice([-1, 1]) * rng.uniform(0.02, 0.06) + 0.5) % 1.0 - 0.5
        m = (rng.random(C) > 0.1).astype(np.uint8)
        if m.sum() < 4:
            m[:4] = 1
        data.append(inp["data"]); x0.append(x); errs.append(inp["errs"]); masks.append(m)
        nuf.append([g["nu_fit"]] * 3); Ps.append(inp["P"])
    data, x0, errs, masks, nuf, Ps = map(np.array, (data, x0, errs, masks, nuf, Ps))
    for method in ("trust-ncg", "newton"):
        for seed_ns in (0, 64):
            if seed_ns and scat:
                continue
            kw = dict(errs=errs, chan_mask=masks, nu_fits=nuf, nu_outs=nuf, fit_flags=flags, log10_tau=l10,
                      method=method, seed_ns=seed_ns)
            whole = eng.fit_batch(data, freqs, Ps, x0, **kw)
            # the same batch through the device in sub-batches of two or three subints
            eng.set_option("max_work_bytes", 2.7 * C * nbin * 24)
            try:
                parts = eng.fit_batch(data, freqs, Ps, x0, **kw)
            finally:
                eng.set_option("max_work_bytes", 96e9)
            dsub = np.abs(whole["params"] - parts["params"]).max()
            wsub = worst.setdefault(("sub-batches", method, seed_ns), [0.0, 0, 0])
            wsub[0] = max(wsub[0], dsub); wsub[1] += int((whole["nfeval"] != parts["nfeval"]).sum()); wsub[2] += N
            for i in range(N):
                one = eng.fit_batch(data[i:i + 1], freqs, Ps[i:i + 1], x0[i:i + 1], errs=errs[i:i + 1],
                                    chan_mask=masks[i:i + 1], nu_fits=nuf[i:i + 1], nu_outs=nuf[i:i + 1],
                                    fit_flags=flags, log10_tau=l10, method=method, seed_ns=seed_ns)
                d = np.abs(whole["params"][i] - one["params"][0])
                d[0] = min(d[0], abs(d[0] - 1.0))
                key = ("".join(map(str, flags)), method, seed_ns)
                w = worst.setdefault(key, [0.0, 0, 0])
                w[0] = max(w[0], d.max()); w[1] += int(whole["nfeval"][i] != one["nfeval"][0]); w[2] += 1
                if d.max() > 1e-12:
                    print("  batch %d subint %d %s %s seed %d: max |dparam| %.2e nfev %d vs %d rc %d vs %d" % (
                        b, i, key[0], method, seed_ns, d.max(), whole["nfeval"][i], one["nfeval"][0],
                        whole["return_code"][i], one["return_code"][0]))
                nfit += 1
print("%d fits compared" % nfit)
for key, w in sorted(worst.items()):
    print("  %s %-9s seed_ns %2d: max |dparam| %.2e, evaluation counts differ in %d of %d" % (key[0], key[1], key[2], w[0], w[1], w[2]))
