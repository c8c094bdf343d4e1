"""Linear-tau trust-ncg fits of configs[3] shape: the subints where the model path and
the ordinary path count different evaluations, against the CPU oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case
from oracle import pptoas_oracle as orc

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flags = [1, 1, 0, 1, 1]
e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(2048, 2048, flags, False, nsub=nsub, tau_us=20.0)
res = {}
for sm in (0, 1):
    e.set_option("scat_model", sm)
    res[sm] = e.fit_batch(data, freqs, P, x0, per_channel=False, **kw)
bad = np.where(res[0]["nfeval"] != res[1]["nfeval"])[0]
print("mismatching subints:", bad, res[0]["nfeval"][bad], res[1]["nfeval"][bad])
for i in list(bad[:2]) + [0]:
    host = data[i].cpu().numpy()
    o = orc.fit_portrait_full(host, model, x0[i], P[i], freqs, [nu_fit] * 3, [None] * 3, errs[i], flags, log10_tau=False)
    print("subint", i, "oracle nfev", o.nfeval, "params", np.asarray(o.params), "nu", o.nu_DM, o.nu_tau)
    for sm in (0, 1):
        r = res[sm]
        print("   model=%d nfev %d  dparams vs oracle %s  nu %s" % (
            sm, r["nfeval"][i], np.array2string(r["params"][i] - np.asarray(o.params), precision=3), r["nu_refs"][i]))
