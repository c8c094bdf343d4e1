"""Scattering model of the closing iterations against the ordinary path (GPU box):
  python tools/dev_scatmodel.py [nsub]
Prints parameter differences, evaluation counts and kernel-family times of
configs[3]-shaped fits with the model on and off, for both minimisers."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for l10 in (True, False):
    e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(
        2048, 2048, [1, 1, 0, 1, 1], l10, nsub=nsub, tau_us=20.0)
    for method in ("trust-ncg", "newton"):
        res = {}
        for sm in (0, 1):
            e.set_option("scat_model", sm)
            e.set_option("profile", 0)
            e.fit_batch(data, freqs, P, x0, method=method, per_channel=False, **kw)
            e.set_option("profile", 1)
            e.kernel_times(reset=True)
            t0 = time.perf_counter()
            r = e.fit_batch(data, freqs, P, x0, method=method, per_channel=False, **kw)
            dt = time.perf_counter() - t0
            kt = e.kernel_times(reset=True)
            res[sm] = r
            print("log10=%d %s model=%d: %.2f ms  nfev mean %.2f  rc %s  " % (
                l10, method, sm, dt * 1e3, r["nfeval"].mean(), np.unique(r["return_code"])),
                {k: (round(v[0] * 1e3, 3), v[1]) for k, v in kt.items() if v[1]})
        a, b = res[0], res[1]
        d = np.abs(a["params"] - b["params"]).max(axis=0)
        print("   max |dparams|", d, " nfev equal:", (a["nfeval"] == b["nfeval"]).mean(),
              " chi2 rel", np.abs(a["chi2"] / b["chi2"] - 1).max(),
              " errs rel", np.nanmax(np.abs(a["param_errs"][:, [0, 1, 3, 4]] / b["param_errs"][:, [0, 1, 3, 4]] - 1)))
