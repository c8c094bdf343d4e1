"""GPU box: the same batch fitted with an engine option off and on; largest difference of
every output field (python tools/dev_ab_kernels.py one_exchange [nsub] [f32])."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pulseportraiture_amd.engine import Engine
from pulseportraiture_amd import gmodel
from pulseportraiture_amd.pplib import guess_fit_freq, Dconst

opt = sys.argv[1] if len(sys.argv) > 1 else "one_exchange"
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
C, B = 4096, 2048
e = Engine(0)
freqs, model, P0 = gmodel.example_model(C, B)
e.set_model(model)
rng = np.random.default_rng(7)
P = np.full(nsub, P0)
inj = np.zeros((nsub, 3))
inj[:, 0] = rng.uniform(-0.5, 0.5, nsub)
inj[:, 1] = 34.56789 + rng.normal(3e-4, 2e-4, nsub)
data = torch.empty((nsub, C, B), dtype=dt, device="cuda:0")
e.synth_portraits(data, freqs, P, inj, 0.05, 20260101, 0)
nu_fit = float(guess_fit_freq(freqs))
x0 = np.zeros((nsub, 5))
x0[:, 0] = (inj[:, 0] + Dconst * inj[:, 1] / P / nu_fit ** 2 + 1e-4 * rng.standard_normal(nsub) + 0.5) % 1.0 - 0.5
x0[:, 1] = 34.56789
kw = dict(errs=np.full((nsub, C), 0.05), nu_fits=np.full((nsub, 3), nu_fit), fit_flags=[1, 1, 0, 0, 0])
res = {}
for v in (0, 1):
    e.set_option(opt, v)
    res[v] = e.fit_batch(data, freqs, P, x0, **kw)
a, b = res[0], res[1]
for k in a:
    if isinstance(a[k], np.ndarray) and a[k].dtype.kind == "f" and a[k].shape == b[k].shape:
        d = np.abs(a[k] - b[k])
        sc = np.maximum(np.abs(a[k]), 1e-300)
        print("%-14s max |diff| %.3e   max rel %.3e" % (k, np.nanmax(d), np.nanmax(d / sc)))
print("phase errs (rot): median", np.median(a["param_errs"][:, 0]))
for j in range(2):
    print("param %d: max |diff| %.3e" % (j, np.max(np.abs(a["params"][:, j] - b["params"][:, j]))))
# third flow: moments from the stored cross-spectrum (k_eval_moments), generic transform kernel
e.set_option(opt, 0)
e.set_option("moments_in_xspec", 0)
c = e.fit_batch(data, freqs, P, x0, **kw)
e.set_option("moments_in_xspec", 1)
for name, r in (("option=0", a), ("option=1", b)):
    print("%s vs stored-X flow: phi %.3e  DM %.3e  scales %.3e" % (
        name, np.max(np.abs(r["params"][:, 0] - c["params"][:, 0])), np.max(np.abs(r["params"][:, 1] - c["params"][:, 1])),
        np.max(np.abs(r["scales"] - c["scales"]))))
from oracle import pptoas_oracle as orc
o = orc.fit_portrait_full(data[0].double().cpu().numpy(), model, x0[0], P[0], freqs, [nu_fit] * 3,
                          [None] * 3, kw["errs"][0], [1, 1, 0, 0, 0], log10_tau=False)
for name, r in (("option=0", a), ("option=1", b), ("stored-X", c)):
    print("%s vs oracle (subint 0): phi %.3e  DM %.3e  scales %.3e nu_DM %.3e" % (
        name, abs(r["params"][0, 0] - o.phi), abs(r["params"][0, 1] - o.DM), np.max(np.abs(r["scales"][0] - o.scales)),
        abs(r["nu_refs"][0, 0] - o.nu_DM)))
w = int(np.argmax(np.abs(a["params"][:, 1] - b["params"][:, 1]) / a["param_errs"][:, 1] + np.abs(a["params"][:, 0] - b["params"][:, 0]) / a["param_errs"][:, 0]))
o = orc.fit_portrait_full(data[w].double().cpu().numpy(), model, x0[w], P[w], freqs, [nu_fit] * 3,
                          [None] * 3, kw["errs"][w], [1, 1, 0, 0, 0], log10_tau=False)
print("worst subint", w, "nfeval", a["nfeval"][w], b["nfeval"][w], c["nfeval"][w], "oracle nfev", o.nfeval)
for name, r in (("option=0", a), ("option=1", b), ("stored-X", c)):
    print("%s vs oracle (subint %d): phi %.3e  DM %.3e  scales %.3e nu_DM %.3e  (errs %.2e %.2e)" % (
        name, w, abs(r["params"][w, 0] - o.phi), abs(r["params"][w, 1] - o.DM), np.max(np.abs(r["scales"][w] - o.scales)),
        abs(r["nu_refs"][w, 0] - o.nu_DM), r["param_errs"][w, 0], r["param_errs"][w, 1]))
