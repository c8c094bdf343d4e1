"""Iterates of the trust-ncg scattering fit of one configs[3]-shaped subint (GPU box):
distance of every accepted point from the final one, in the per-channel units the
scattering model cares about (max |dphi_n| [rot], max |dtau_n/tau_n|)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case
from pulseportraiture_amd.pplib import Dconst

l10 = True
e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(2048, 2048, [1, 1, 0, 1, 1], l10, nsub=4, tau_us=20.0)
e.set_option("scat_model", 0)
its = []
for k in range(1, 20):
    e.set_option("max_iter", k)
    r = e.fit_batch(data, freqs, P, x0, per_channel=False, **dict(kw, nu_outs=np.full((4, 3), nu_fit)))
    its.append((r["params"][0].copy(), int(r["nfeval"][0]), int(r["return_code"][0])))
fin = its[-1][0]
p1 = Dconst * (freqs ** -2 - nu_fit ** -2) / P[0]
lnf = np.log(freqs / nu_fit)
for k, (x, nfev, rc) in enumerate(its):
    d = x - fin
    dphi = np.abs(d[0] + d[1] * p1).max()
    rel = np.abs(np.log(10) * d[3] + d[4] * lnf).max()
    print("max_iter %2d nfev %2d rc %d  dphi_n %.2e  rel tau_n %.2e   x-fin %s" % (k + 1, nfev, rc, dphi, rel, np.array2string(d, precision=2)))
