"""Rounding noise of the gradient at a converged scattering fit: device evaluator and
NumPy oracle against an 80-bit evaluation of the same sums.  (GPU box)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _full_shape_case
from oracle import pptoas_oracle as orc

ld, cld = np.longdouble, np.clongdouble
PI = ld("3.14159265358979323846264338327950288")


def grad_ld(params, dFT, mFT, errs_FT, P, freqs, nuDM, nutau, l10):
    phi, DM, GM, tau, alpha = [ld(v) for v in params]
    t10 = tau
    if l10:
        tau = ld(10) ** tau
    C, H = dFT.shape
    k = np.arange(H).astype(ld)
    f = freqs.astype(ld)
    Dc = ld("0.000241") ** -1
    p1 = Dc * (f ** -2 - ld(nuDM) ** -2) / ld(P)
    phis = phi + DM * p1
    lnf = np.log(f / ld(nutau))
    taus = tau * np.exp(alpha * lnf)
    fr = np.outer(phis, k)
    fr = fr - np.rint(fr)
    ph = np.cos(2 * PI * fr) + 1j * np.sin(2 * PI * fr)
    kap = 2 * PI * k
    B = 1 / (1 + 1j * np.outer(taus, kap))
    z = dFT.astype(cld) * np.conj(mFT.astype(cld)) * ph
    M = np.abs(mFT.astype(cld)) ** 2
    w = errs_FT.astype(ld) ** -2
    zb = z * np.conj(B)
    A0 = zb.real.sum(-1) * w
    A1 = (1j * kap * zb).real.sum(-1) * w
    Bt = -1j * kap * B ** 2
    T1 = (z * np.conj(Bt)).real.sum(-1) * w
    S0 = (np.abs(B) ** 2 * M).sum(-1) * w
    S1 = (2 * (B * np.conj(Bt)).real * M).sum(-1) * w
    q1 = np.log(ld(10)) * taus if l10 else taus / tau
    q2 = lnf * taus
    dC = [A1, A1 * p1, None, T1 * q1, T1 * q2]
    dS = [0 * S0, 0 * S0, None, S1 * q1, S1 * q2]
    g = np.zeros(5, dtype=ld)
    for j in (0, 1, 3, 4):
        g[j] = -((A0 ** 2 / S0) * (2 * dC[j] / A0 - dS[j] / S0)).sum()
    f0 = -(A0 ** 2 / S0).sum()
    return f0, g


if __name__ == '__main__':
    flags, l10 = [1, 1, 0, 1, 1], True
    nsub = 4
    e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(256, 1024, flags, l10, nsub=nsub, tau_us=30.0, seed=9)
    e.set_option("scat_model", 0)
    r = e.fit_batch(data, freqs, P, x0, nu_outs=np.full((nsub, 3), nu_fit), method='newton', **kw)
    xs = r["params"].copy()
    ro = e.fit_batch(data, freqs, P, xs, objective=True, nu_outs=np.full((nsub, 3), nu_fit), **dict(kw))
    host = data.cpu().numpy()
    mFT = np.fft.rfft(model, axis=-1); mFT[:, 0] = 0
    for i in range(nsub):
        dFT = np.fft.rfft(host[i], axis=-1); dFT[:, 0] = 0
        eFT = errs[i] * np.sqrt(1024 / 2.0)
        args = (dFT, mFT, eFT, P[i], freqs, nu_fit, nu_fit, nu_fit, flags, l10)
        go = orc.fit_portrait_full_function_deriv(xs[i], *args)
        fo = orc.fit_portrait_full_function(xs[i], *args)
        fl, gl = grad_ld(xs[i], dFT, mFT, eFT, P[i], freqs, nu_fit, nu_fit, l10)
        gd = ro["obj_grad"][i] if "obj_grad" in ro else ro["g0"][i]
        fd = ro["obj_f"][i] if "obj_f" in ro else ro["f0"][i]
        print("subint", i, " f %.6e  ulp(f) %.2e" % (float(fl), np.spacing(abs(float(fl)))))
        print("   f err: device %.2e  numpy %.2e" % (float(ld(fd) - fl), float(ld(fo) - fl)))
        print("   g 80-bit      ", np.array2string(gl.astype(float), precision=3))
        print("   g err device  ", np.array2string((gd.astype(ld) - gl).astype(float), precision=3))
        print("   g err numpy   ", np.array2string((go.astype(ld) - gl).astype(float), precision=3))

