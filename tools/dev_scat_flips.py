"""Where the model path and the ordinary path count different evaluations: how far
apart do they end?  (GPU box)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case, _dphi_arr

for l10, flags in [(True, [1, 0, 0, 1, 1]), (False, [1, 1, 0, 1, 0]), (True, [1, 1, 0, 1, 1]), (True, [1, 1, 0, 1, 0])]:
    e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(256, 1024, flags, l10, nsub=64, tau_us=30.0, seed=9)
    res = {}
    for sm in (0, 1):
        e.set_option("scat_model", sm)
        res[sm] = e.fit_batch(data, freqs, P, x0, nu_outs=np.full((64, 3), nu_fit), **kw)
    a, b = res[0], res[1]
    ne = a["nfeval"] != b["nfeval"]
    print(l10, flags, "flips %d/64" % ne.sum())
    for i in np.where(ne)[0][:8]:
        d = a["params"][i] - b["params"][i]
        print("   nfev %2d %2d  dphi %.2e dDM %.2e dtau %.2e dalpha %.2e   (errs %.1e %.1e %.1e %.1e) dchi2 %.2e" % (
            a["nfeval"][i], b["nfeval"][i], _dphi_arr(a["params"][i, 0], b["params"][i, 0]), d[1], d[3], d[4],
            a["param_errs"][i, 0], a["param_errs"][i, 1], a["param_errs"][i, 3], a["param_errs"][i, 4],
            a["chi2"][i] - b["chi2"][i]))
    eq = ~ne
    print("   equal-count subints: max dphi %.2e dtau %.2e" % (_dphi_arr(a["params"][eq, 0], b["params"][eq, 0]).max(),
          np.abs(a["params"][eq, 3] - b["params"][eq, 3]).max()))
