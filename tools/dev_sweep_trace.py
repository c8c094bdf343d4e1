"""One case of tools/sweep_parity.py iteration by iteration: SciPy's own loop (its
CGSteihaugSubproblem, the oracle's f/g/H) beside the device's trace (library built
with -DPP_STEP_TRACE=0).  (GPU box)   python tools/dev_sweep_trace.py <case>"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tools.sweep_parity import make_case
from oracle import pptoas_oracle as orc
from scipy.optimize._trustregion_ncg import CGSteihaugSubproblem
from pulseportraiture_amd.engine import Engine

c = make_case(int(sys.argv[1]))
print("case", c["k"], c["flags"], "log10", c["l10"], "C", c["C"], "nbin", c["nbin"], "option", c["option"])
eng = Engine(0)
eng.set_option("scat_model", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng.set_model(c["model"])
kw = dict(errs=c["errs"][None], chan_mask=c["mask"][None], nu_fits=[[c["nu_fit"]] * 3],
          nu_outs=[[c["nu_fit"]] * 3], fit_flags=c["flags"], log10_tau=c["l10"], option=c["option"])
eng.set_option("taylor", 0)
r = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], **kw)
print("device nfev", r["nfeval"][0], "params", r["params"][0])
ok = np.where(c["mask"])[0]
B = c["nbin"]
dFT = np.fft.rfft(c["data"][ok], axis=-1); dFT[:, 0] = 0
mFT = np.fft.rfft(c["model"][ok], axis=-1); mFT[:, 0] = 0
eFT = c["errs"][ok] * np.sqrt(B / 2.0)
args = (dFT, mFT, eFT, c["P"], c["freqs"][ok], c["nu_fit"], c["nu_fit"], c["nu_fit"], [bool(f) for f in c["flags"]], c["l10"])
fun = lambda x: orc.fit_portrait_full_function(x, *args)
jac = lambda x: orc.fit_portrait_full_function_deriv(x, *args)
hess = lambda x: orc.fit_portrait_full_function_2deriv(x, *args)
x = np.asarray(c["x0"], dtype=float)
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
        print("scipy exit: pred", pred, "after", k, "iterations; x =", x)
        break
    rho = actual / pred
    print("np  it %2d f %.17g f_new %.17g actual %.3e pred %.3e rho %.3f radius %.3e hits %d |p| %.3e" % (
        k, m.fun, mp.fun, actual, pred, rho, radius, hits, np.linalg.norm(p)))
    if rho < 0.25:
        radius *= 0.25
    elif rho > 0.75 and hits:
        radius = min(2 * radius, 1000.0)
    if rho > 0.15:
        x, m = xp, mp
    k += 1
    if k > 60:
        break
