"""Poor guesses: whatever route a fit takes (certified in one pass, one re-expansion,
evaluations over a stored cross-spectrum), it must end where the fit from a good guess
ends.  Random (phi, DM[, GM]) problems with DM guesses off by up to 1e-2 pc cm^-3 and
phase guesses off by up to 0.05 rot, both solvers, against the Newton answer from the
caller-quality guess.  (GPU box)   python tools/sweep_poor_guess.py [n]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.synth_host import make_inputs, caller_guess, model_portrait
from pulseportraiture_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = Engine(0)
rng = np.random.default_rng(99)
stats = {}
for k in range(n):
    flags = [[1, 1, 0, 0, 0], [1, 1, 1, 0, 0], [1, 0, 0, 0, 0]][k % 3]
    C = int(rng.integers(8, 64)); nbin = int(2 ** rng.integers(7, 12))
    freqs, model = model_portrait(C, nbin)
    eng.set_model(model)
    inp = make_inputs(C, nbin, 52000 + k, model=model, DM0=(34.56789 if rng.random() < 0.3 else 0.0),
                      sigma=float(rng.choice([0.03, 0.1])), GM=(0.25 if flags[2] else None))
    g = caller_guess(inp)
    kw = dict(errs=inp["errs"][None], nu_fits=[[g["nu_fit"]] * 3], nu_outs=[[g["nu_fit"]] * 3], fit_flags=flags)
    ref = eng.fit_batch(inp["data"][None], freqs, inp["P"], g["init_params"], method='newton', **kw)
    x = g["init_params"].copy()
    kind = ["dm", "phase", "both"][int(rng.integers(0, 3))]
    if kind in ("dm", "both") and flags[1]:
        x[1] += rng.choice([-1, 1]) * 10 ** rng.uniform(-3.3, -2.0)
    if kind in ("phase", "both"):
        x[0] = (x[0] + rng.choice([-1, 1]) * 10 ** rng.uniform(-3.0, -1.3) + 0.5) % 1.0 - 0.5
    for method in ("trust-ncg", "newton"):
        r = eng.fit_batch(inp["data"][None], freqs, inp["P"], x, method=method, **kw)
        d = np.abs(r["params"][0] - ref["params"][0]); d[0] = min(d[0], abs(d[0] - 1.0))
        sig = np.where(ref["param_errs"][0] > 0, ref["param_errs"][0], 1.0)
        key = ("".join(map(str, flags)), method)
        s = stats.setdefault(key, dict(n=0, dphi=[], dsig=[], nfev=[], bad=0))
        s["n"] += 1; s["dphi"].append(d[0]); s["dsig"].append((d / sig).max()); s["nfev"].append(int(r["nfeval"][0]))
        if d[0] > 5e-9 or int(r["return_code"][0]) != 2:
            s["bad"] += 1
            print("  case %d %s %s %s: dphi %.2e dsig %.2e nfev %d rc %d" % (k, key[0], method, kind, d[0], (d / sig).max(),
                                                                          r["nfeval"][0], r["return_code"][0]))
for key, s in sorted(stats.items()):
    nf = np.array(s["nfev"])
    print("%s %-9s n=%3d  |dphi| median %.1e max %.1e  max |dparam|/sigma %.1e  passes: 1 %d, 2 %d, more %d   bad %d" % (
        key[0], key[1], s["n"], np.median(s["dphi"]), max(s["dphi"]), max(s["dsig"]), (nf == 1).sum(), (nf == 2).sum(),
        (nf > 2).sum(), s["bad"]))
