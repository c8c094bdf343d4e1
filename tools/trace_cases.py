"""Per-iteration traces of single sweep cases (tools/sweep_parity.py's make_case): the device's
trust-ncg walk (library built with -DPP_TAYLOR_TRACE=3 -DPP_STEP_TRACE=0, e.g.
`tools/build_variants.sh trace "-DPP_TAYLOR_TRACE=3 -DPP_STEP_TRACE=0"` and PP_TOAS_LIB=variants/trace.so)
beside SciPy's own loop driven with the oracle's f / g / H -- to find the first decision where a fit
that misses the reference's raw answer parts from it.  (GPU box)
    PP_TOAS_LIB=variants/trace.so python tools/trace_cases.py 2212 2362 ..."""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from tools.sweep_parity import make_case
from oracle import pptoas_oracle as orc
from scipy.optimize._trustregion_ncg import CGSteihaugSubproblem


def scipy_trace(c):
    ok = np.where(c["mask"])[0]
    data, model, freqs, errs = c["data"][ok], c["model"][ok], c["freqs"][ok], c["errs"][ok]
    nbin = data.shape[-1]
    mFT = np.fft.rfft(model, axis=-1); mFT[:, 0] = 0
    dFT = np.fft.rfft(data, axis=-1); dFT[:, 0] = 0
    eFT = errs * np.sqrt(nbin / 2.0)
    nu = c["nu_fit"]
    args = (dFT, mFT, eFT, c["P"], freqs, nu, nu, nu, [bool(f) for f in c["flags"]], c["l10"])
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
            print("np  exit: pred %.3e after %d iterations; x = %s" % (pred, k, np.array2string(x, precision=17)))
            break
        rho = actual / pred
        print("np  it %2d f %.17g f_new %.17g actual %.3e pred %.3e rho %.3f radius %.3e hits %d |p| %.3e same %d" % (
            k, m.fun, mp.fun, actual, pred, rho, radius, hits, np.linalg.norm(p), int(np.array_equal(xp, x))))
        if rho < 0.25:
            radius *= 0.25
        elif rho > 0.75 and hits:
            radius = min(2 * radius, 1000.0)
        if rho > 0.15:
            x, m = xp, mp
        k += 1
        if k > 80:
            break
    return x


if __name__ == "__main__":
    from pulseportraiture_amd.engine import Engine
    eng = Engine(0)
    for kv in os.environ.get("PP_SWEEP_OPTS", "").split():
        name, _, val = kv.partition("=")
        eng.set_option(name, float(val))
    for k in (int(v) for v in sys.argv[1:]):
        c = make_case(k)
        print("==== case %d flags %s l10 %d C %d (used %d) nbin %d x0 %s" % (
            k, c["flags"], c["l10"], c["C"], int(c["mask"].sum()), c["nbin"], np.array2string(np.asarray(c["x0"]), precision=17)))
        sys.stdout.flush()
        eng.set_model(c["model"])
        kw = dict(errs=c["errs"][None], chan_mask=c["mask"][None], nu_fits=[[c["nu_fit"]] * 3],
                  nu_outs=[[c["nu_fit"]] * 3], fit_flags=c["flags"], log10_tau=c["l10"], option=c["option"])
        r = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], **kw)
        eng.synchronize()
        sys.stdout.flush()
        print("dev result nfev %d rc %d params %s" % (r["nfeval"][0], r["return_code"][0],
                                                        np.array2string(r["params"][0], precision=17)))
        xs = scipy_trace(c)
        d = r["params"][0] - xs
        d[0] = (d[0] + 0.5) % 1.0 - 0.5
        print("dev - scipy-loop: %s" % np.array2string(d, precision=3))
        sys.stdout.flush()
