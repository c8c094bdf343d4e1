"""How often does SciPy's own iteration (the CPU oracle) run into the tail of rejected
steps, against the device's two paths?  (GPU box; 24 subints of 256 x 1024)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case, _dphi_arr
from oracle import pptoas_oracle as orc

for l10, flags in [(False, [1, 1, 0, 1, 0]), (True, [1, 0, 0, 1, 1]), (True, [1, 1, 0, 1, 1])]:
    nsub = 24
    e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(256, 1024, flags, l10, nsub=nsub, tau_us=30.0, seed=9)
    res = {}
    for sm in (0, 1):
        e.set_option("scat_model", sm)
        res[sm] = e.fit_batch(data, freqs, P, x0, nu_outs=np.full((nsub, 3), nu_fit), **kw)
    host = data.cpu().numpy()
    on, dphi0, dphi1 = [], [], []
    for i in range(nsub):
        o = orc.fit_portrait_full(host[i], model, x0[i], P[i], freqs, [nu_fit] * 3, [nu_fit] * 3, errs[i], flags, log10_tau=l10)
        on.append(o.nfeval)
        dphi0.append(_dphi_arr(res[0]["params"][i, 0], o.phi)); dphi1.append(_dphi_arr(res[1]["params"][i, 0], o.phi))
    print(l10, flags)
    print("  oracle nfev-1 :", np.array(on) - 1)
    print("  device plain  :", res[0]["nfeval"])
    print("  device model  :", res[1]["nfeval"])
    print("  |dphi| vs oracle, plain: max %.2e median %.2e ; model: max %.2e median %.2e" % (
        max(dphi0), np.median(dphi0), max(dphi1), np.median(dphi1)))
