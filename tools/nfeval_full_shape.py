#!/usr/bin/env python3
"""nfeval at the stated shapes, device against oracle (run on the GPU box):  python tools/nfeval_full_shape.py [out]

The subints of tests/test_gpu_parity.py::test_headline_shape_matches_oracle_raw (4096 x 2048, phase + DM, f64 and f32
portraits: 5 + 5) and ::test_full_shapes_of_cfg3_and_cfg4_match_oracle (4096 x 2048 + GM: 3; 2048 x 2048 scattering: 3):
the device's evaluation count (SciPy's nfev, counted by SciPy's rule) beside the oracle's.  Where they differ, the
ORACLE is refitted with its channels in other orders -- reversed and 14 seeded permutations; nothing changes but the
order in which NumPy adds the per-channel terms -- and the counts it then reports are listed: the last unit of nfeval
hangs on whether the closing proposal p = -H^-1 g (~1e-15) rounds to x itself, i.e. on the rounding noise of g of
whoever computes it, so the oracle's own count moves with the order.  Where no channel order moves it
(the noise that decides sits in the sums over HARMONICS, which a channel permutation does not reorder -- as for case
194, profiles/r05_ref_exit_other_cpu.txt), the oracle is refitted in child interpreters whose NumPy has its SIMD
dispatch narrowed (NPY_DISABLE_CPU_FEATURES: AVX512 off / AVX2 + FMA3 off as well -- the same code on an older host),
six channel orders each."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pptoas_oracle as orc          # noqa: E402
from tests import test_gpu_parity as T          # noqa: E402


FEATURE_SETS = [("as built", ""),
                ("AVX512 off", "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX512_KNL AVX512_KNM"),
                ("AVX2 / FMA3 / AVX512 off", "AVX2 FMA3 AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX512_KNL AVX512_KNM")]
CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import pptoas_oracle as orc
z = np.load(%(npz)r)
C = z["host"].shape[0]
orders = [np.arange(C), np.arange(C)[::-1]] + [np.random.default_rng(200 + k).permutation(C) for k in range(4)]
res = []
for pm in orders:
    o = orc.fit_portrait_full(z["host"][pm], z["model"][pm], z["x0"], float(z["P"]), z["freqs"][pm], [float(z["nu_fit"])] * 3,
                              [None] * 3, z["errs"][pm], [int(v) for v in z["flags"]], log10_tau=bool(z["l10"]))
    res.append([int(o.nfeval), float(o.phi)])
print(json.dumps(res))
"""


def other_cpus(host, model, x0, P, freqs, nu_fit, errs, flags, l10):
    import json
    import subprocess
    import tempfile
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        npz = os.path.join(tmp, "case.npz")
        np.savez(npz, host=host, model=model, x0=x0, P=P, freqs=freqs, nu_fit=nu_fit, errs=errs, flags=np.asarray(flags), l10=l10)
        for label, disable in FEATURE_SETS:
            env = dict(os.environ, OMP_NUM_THREADS="1")
            if disable:
                env["NPY_DISABLE_CPU_FEATURES"] = disable
            r = subprocess.run([sys.executable, "-W", "ignore", "-c", CHILD % dict(root=ROOT, npz=npz)], env=env, capture_output=True, text=True)
            try:
                rows.append((label, json.loads(r.stdout.strip().splitlines()[-1])))
            except Exception:
                rows.append((label, "failed: " + r.stderr[-200:]))
    return rows


def scipy_walk(host, model, x0, P, freqs, nu_fit, errs, flags, l10, out):
    """SciPy's own trust-ncg loop (scipy/optimize/_trustregion.py with CGSteihaugSubproblem) on the oracle's f, g, H,
    printed iteration by iteration: the proposal's length against the spacing of the doubles at x -- the last unit of
    nfev is whether fl(x + p) is x itself (SciPy's one-point cache answers, not counted) -- and the predicted reduction
    against ulp(f)."""
    from scipy.optimize._trustregion_ncg import CGSteihaugSubproblem
    B = host.shape[1]
    mFT = np.fft.rfft(model, axis=-1); mFT[:, 0] = 0
    dFT = np.fft.rfft(host, axis=-1); dFT[:, 0] = 0
    args = (dFT, mFT, errs * np.sqrt(B / 2.0), P, freqs, nu_fit, nu_fit, nu_fit, [bool(f) for f in flags], l10)
    fun = lambda x: orc.fit_portrait_full_function(x, *args)
    jac = lambda x: orc.fit_portrait_full_function_deriv(x, *args)
    hess = lambda x: orc.fit_portrait_full_function_2deriv(x, *args)
    x = np.asarray(x0, dtype=float).copy()
    radius, k, nfev, last = 1.0, 0, 1, x.copy()
    m = CGSteihaugSubproblem(x, fun, jac, hess, None)
    ii = np.where(flags)[0]
    while k < 60:
        pstep, hits = m.solve(radius)
        pv = m(pstep)
        xp = x + pstep
        same = bool(np.array_equal(xp, last))
        if not same:
            nfev += 1
            last = xp.copy()
        mp = CGSteihaugSubproblem(xp, fun, jac, hess, None)
        actual, pred = m.fun - mp.fun, m.fun - pv
        ulpf = np.spacing(abs(m.fun))
        print("%-24s        scipy it %2d  |p|/spacing(x) %s  pred %.3g (%.2f ulp f)  actual %.3g (%.2f ulp f)  fl(x + p) == x: %s  nfev %d" %
              ("", k, " ".join("%.3g" % (abs(pstep[j]) / np.spacing(abs(x[j]))) for j in ii), pred, pred / ulpf, actual, actual / ulpf, same, nfev), file=out)
        if pred <= 0:
            break
        rho = actual / pred
        if rho < 0.25:
            radius *= 0.25
        elif rho > 0.75 and hits:
            radius = min(2 * radius, 1000.0)
        if rho > 0.15:
            x = xp
            m = mp
        k += 1


def main():
    import torch
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    cases = [("headline-f64", 4096, 2048, [1, 1, 0, 0, 0], False, None, False, 5, 17, "f64"),
             ("headline-f32", 4096, 2048, [1, 1, 0, 0, 0], False, None, False, 5, 17, "f32"),
             ("cfg3-4096x2048-phiDMGM", 4096, 2048, [1, 1, 1, 0, 0], False, None, True, 3, 5, "f64"),
             ("cfg4-2048x2048-scat", 2048, 2048, [1, 1, 0, 1, 1], True, 20.0, False, 3, 5, "f64")]
    print("%-24s %4s %7s %7s   %s" % ("case", "sub", "device", "oracle", "oracle under other channel orders (count: orders) / device's answer among them"), file=out)
    for name, C, B, flags, l10, tau_us, gm, nsub, seed, dt in cases:
        e, data, freqs, model, P, x0, errs, nu_fit, kw = T._full_shape_case(C, B, flags, l10, nsub=nsub, tau_us=tau_us, gm=gm, seed=seed)
        if dt == "f32":
            data = data.to(torch.float32)
        r = e.fit_batch(data, freqs, P, x0, **kw)
        for i in range(nsub):
            host = data[i].cpu().numpy().astype(np.float64)
            o = orc.fit_portrait_full(host, model, x0[i], P[i], freqs, [nu_fit] * 3, [None] * 3, errs[i], flags, log10_tau=l10)
            dev = int(r["nfeval"][i])
            note = ""
            if dev != o.nfeval:
                counts, hit = {}, 0
                perms = [np.arange(C)[::-1]] + [np.random.default_rng(100 + k).permutation(C) for k in range(14)]
                for pm in perms:
                    op = orc.fit_portrait_full(host[pm], model[pm], x0[i], P[i], freqs[pm], [nu_fit] * 3, [None] * 3,
                                               errs[i][pm], flags, log10_tau=l10)
                    counts[op.nfeval] = counts.get(op.nfeval, 0) + 1
                    if op.nfeval == dev and abs((op.phi - r["params"][i, 0] + 0.5) % 1 - 0.5) < 1e-12:
                        hit += 1
                note = "  ".join("%d: %d" % kv for kv in sorted(counts.items())) + \
                    "   / device's count AND answer (|dphi| < 1e-12) under %d of %d orders" % (hit, len(perms))
            print("%-24s %4d %7d %7d   %s" % (name, i, dev, o.nfeval, note), file=out)
            if dev != o.nfeval:
                scipy_walk(host, model, x0[i], P[i], freqs, nu_fit, errs[i], flags, l10, out)
            if dev != o.nfeval and dev not in counts:
                for label, res in other_cpus(host, model, x0[i], P[i], freqs, nu_fit, errs[i], flags, l10):
                    if isinstance(res, str):
                        print("%-24s        NumPy SIMD %-26s %s" % ("", label, res), file=out)
                        continue
                    cnt = {}
                    for nf, ph in res:
                        cnt[nf] = cnt.get(nf, 0) + 1
                    hit = sum(1 for nf, ph in res if nf == dev)
                    print("%-24s        NumPy SIMD %-26s counts %s  -> the device's count under %d of %d orders" %
                          ("", label, "  ".join("%d: %d" % kv for kv in sorted(cnt.items())), hit, len(res)), file=out)
            out.flush()
        e.close()


if __name__ == "__main__":
    main()
