"""One subint where the device's trust-ncg runs into the tail of rejected steps and
SciPy's does not: SciPy's own loop (its CGSteihaugSubproblem, the oracle's f/g/H) with
every iteration printed, beside the device's trace (library built with
-DPP_STEP_TRACE=<subint>).  (GPU box)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case
from oracle import pptoas_oracle as orc
from scipy.optimize._trustregion_ncg import CGSteihaugSubproblem

isub = int(sys.argv[1]) if len(sys.argv) > 1 else 4
flags, l10 = [1, 0, 0, 1, 1], True
nsub = 24
e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(256, 1024, flags, l10, nsub=nsub, tau_us=30.0, seed=9)
e.set_option("scat_model", 0)
r = e.fit_batch(data, freqs, P, x0, nu_outs=np.full((nsub, 3), nu_fit), **kw)
print("device nfev", r["nfeval"][isub])
host = data[isub].cpu().numpy()
mFT = np.fft.rfft(model, axis=-1); mFT[:, 0] = 0
dFT = np.fft.rfft(host, axis=-1); dFT[:, 0] = 0
eFT = errs[isub] * np.sqrt(1024 / 2.0)
args = (dFT, mFT, eFT, P[isub], freqs, nu_fit, nu_fit, nu_fit, [bool(f) for f in flags], l10)
fun = lambda x: orc.fit_portrait_full_function(x, *args)
jac = lambda x: orc.fit_portrait_full_function_deriv(x, *args)
hess = lambda x: orc.fit_portrait_full_function_2deriv(x, *args)
x = np.asarray(x0[isub], dtype=float)
radius, k = 1.0, 0
m = CGSteihaugSubproblem(x, fun, jac, hess, None)
while True:
    p, hits = m.solve(radius)
    pv = m(p)
    xp = x + p
    mp = CGSteihaugSubproblem(xp, fun, jac, hess, None)
    actual = m.fun - mp.fun
    pred = m.fun - pv
    if pred <= 0:
        print("scipy exit: pred", pred, "after", k, "iterations")
        break
    rho = actual / pred
    print("np  it %2d f %.17g f_new %.17g actual %.3e pred %.3e rho %.3f radius %.3e hits %d" % (k, m.fun, mp.fun, actual, pred, rho, radius, hits))
    if rho < 0.25:
        radius *= 0.25
    elif rho > 0.75 and hits:
        radius = min(2 * radius, 1000.0)
    if rho > 0.15:
        x, m = xp, mp
    k += 1
    if k > 60:
        break
