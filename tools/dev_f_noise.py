"""Scatter of the evaluated objective over a cloud of points within ~1e-9 rot of a
converged scattering fit: device evaluator and NumPy oracle against an 80-bit sum.
(GPU box)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case
from oracle import pptoas_oracle as orc
from tools.dev_grad_noise import grad_ld, ld

flags, l10 = [1, 1, 0, 1, 1], True
e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(256, 1024, flags, l10, nsub=1, tau_us=30.0, seed=9)
e.set_option("scat_model", 0)
r = e.fit_batch(data, freqs, P, x0, nu_outs=np.full((1, 3), nu_fit), method='newton', **kw)
xs = r["params"][0].copy()
npt = 48
rng = np.random.default_rng(3)
pts = np.tile(xs, (npt, 1))
pts[:, 0] += 2e-10 * rng.standard_normal(npt)
pts[:, 3] += 1e-8 * rng.standard_normal(npt)
big = data.expand(npt, -1, -1).contiguous()
kw2 = dict(kw); kw2["errs"] = np.tile(errs, (npt, 1)); kw2["nu_fits"] = np.full((npt, 3), nu_fit)
ro = e.fit_batch(big, freqs, np.full(npt, P[0]), pts, objective=True, nu_outs=np.full((npt, 3), nu_fit), **kw2)
host = data[0].cpu().numpy()
mFT = np.fft.rfft(model, axis=-1); mFT[:, 0] = 0
dFT = np.fft.rfft(host, axis=-1); dFT[:, 0] = 0
eFT = errs[0] * np.sqrt(1024 / 2.0)
args = (dFT, mFT, eFT, P[0], freqs, nu_fit, nu_fit, nu_fit, flags, l10)
ed, en = [], []
for j in range(npt):
    fl, _ = grad_ld(pts[j], dFT, mFT, eFT, P[0], freqs, nu_fit, nu_fit, l10)
    fo = orc.fit_portrait_full_function(pts[j], *args)
    ed.append(float(ld(ro["obj_f"][j]) - fl)); en.append(float(ld(fo) - fl))
ed, en = np.array(ed), np.array(en)
ulp = np.spacing(abs(ro["obj_f"][0]))
print("ulp(f) %.2e" % ulp)
print("device: mean %.2f ulp, scatter (rms about the mean) %.2f ulp, min %.2f max %.2f" % (ed.mean() / ulp, ed.std() / ulp, ed.min() / ulp, ed.max() / ulp))
print("numpy : mean %.2f ulp, scatter %.2f ulp, min %.2f max %.2f" % (en.mean() / ulp, en.std() / ulp, en.min() / ulp, en.max() / ulp))
