"""f32-resident portraits (the paired-split transform kernel for 2048-bin rows, the generic one
otherwise) against the same numbers handed over as f64: the arithmetic is f64 either way, so
the fits must agree to rounding.  Random shapes, masks, families, modes of the transform
(one-pass, stored cross-spectrum, seeded).  (GPU box)   python tools/sweep_f32.py [n]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.synth_host import make_inputs, caller_guess, model_portrait
from pulseportraiture_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = Engine(0)
rng = np.random.default_rng(321)
worst = {}
for k in range(n):
    flags, scat = [([1, 1, 0, 0, 0], False), ([1, 1, 1, 0, 0], False), ([1, 1, 0, 1, 1], True), ([1, 0, 0, 0, 0], False)][k % 4]
    C = int(rng.integers(8, 80)); nbin = int(rng.choice([256, 1024, 2048, 2048, 2048, 4096]))
    N = int(rng.integers(1, 5))
    freqs, model = model_portrait(C, nbin)
    eng.set_model(model)
    data, x0, errs, nuf, Ps = [], [], [], [], []
    for i in range(N):
        tau_us = float(rng.uniform(15, 40)) if scat else None
        inp = make_inputs(C, nbin, 88000 + 10 * k + i, model=model, DM0=(34.56789 if rng.random() < 0.3 else 0.0),
                          sigma=float(rng.choice([0.03, 0.1])), GM=(0.25 if flags[2] else None), tau_us=tau_us)
        g = caller_guess(inp, fit_scat=scat, log10_tau=True, tau_guess_rot=(1.3 * tau_us * 1e-6 / inp["P"]) if scat else None)
        data.append(inp["data"]); x0.append(g["init_params"]); errs.append(inp["errs"]); nuf.append([g["nu_fit"]] * 3); Ps.append(inp["P"])
    d32 = np.array(data).astype(np.float32)
    x0, errs, nuf, Ps = map(np.array, (x0, errs, nuf, Ps))
    m = (rng.random((N, C)) > 0.1).astype(np.uint8)
    for seed_ns in (0, 50):
        if seed_ns and scat:
            continue
        for use_errs in (True, False):
            kw = dict(errs=errs if use_errs else None, chan_mask=m, nu_fits=nuf, nu_outs=nuf, fit_flags=flags,
                      log10_tau=scat, seed_ns=seed_ns, method='newton')
            a = eng.fit_batch(d32, freqs, Ps, x0, **kw)
            b = eng.fit_batch(d32.astype(np.float64), freqs, Ps, x0, **kw)
            d = np.abs(a["params"] - b["params"]); d[:, 0] = np.minimum(d[:, 0], np.abs(d[:, 0] - 1.0))
            sig = np.where(b["param_errs"] > 0, b["param_errs"], 1.0)
            key = (nbin, "".join(map(str, flags)), seed_ns, use_errs)
            w = worst.get(key, 0.0)
            worst[key] = max(w, float((d / sig).max()))
            if (d / sig).max() > 1e-6:
                print("  case %d nbin %d flags %s seed %d errs %s: |dparam|/sigma %.2e  dphi %.2e" % (
                    k, nbin, key[1], seed_ns, use_errs, (d / sig).max(), d[:, 0].max()))
print("max |dparam| / sigma over %d problems: %.2e" % (n, max(worst.values())))
for key in sorted(worst):
    if worst[key] > 1e-9:
        print("  ", key, "%.2e" % worst[key])
