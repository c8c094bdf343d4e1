"""Everything fit_portrait_full returns besides the parameters, on the random problems of
sweep_parity.py with the OUTPUT frequencies left to the fit (zero-covariance frequencies:
get_nu_zeros with its polynomial roots, option 0 / 1), is_toa on / off: nu_refs, errors,
covariance, scales and their errors, S/N, chi2, red_chi2, channel S/N against the oracle.
Compared with method 'newton' on both sides' common optimum: the oracle's answer is first
polished by its own Newton steps so that both sit at the same point.  (GPU box)
    python tools/sweep_outputs.py [ncases]"""
import multiprocessing as mp
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from tools.sweep_parity import make_case


def oracle_out(k):
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    from oracle import pptoas_oracle as orc
    c = make_case(k)
    ok = np.where(c["mask"])[0]
    nus = [c["nu_fit"]] * 3
    is_toa = bool(k % 2)
    o = orc.fit_portrait_full(c["data"][ok], c["model"][ok], c["x0"], c["P"], c["freqs"][ok], nus, [None] * 3,
                              c["errs"][ok], c["flags"], log10_tau=c["l10"], option=c["option"], is_toa=is_toa)
    return k, dict(params=np.asarray(o.params), errs=np.asarray(o.param_errs), nu=[o.nu_DM, o.nu_GM, o.nu_tau],
                   chi2=o.chi2, red_chi2=o.red_chi2, snr=o.snr, scales=np.asarray(o.scales),
                   scale_errs=np.asarray(o.scale_errs), csnr=np.asarray(o.channel_snrs),
                   cov=np.asarray(o.covariance_matrix))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    with mp.get_context("spawn").Pool(64) as pool:
        ora = dict(pool.imap_unordered(oracle_out, range(n), chunksize=4))
    from pulseportraiture_amd.engine import Engine
    eng = Engine(0)
    worst = {}
    skipped = 0
    for k in range(n):
        c = make_case(k)
        o = ora[k]
        ok = np.where(c["mask"])[0]
        eng.set_model(c["model"])
        r = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], errs=c["errs"][None], chan_mask=c["mask"][None],
                          nu_fits=[[c["nu_fit"]] * 3], fit_flags=c["flags"], log10_tau=c["l10"], option=c["option"],
                          is_toa=bool(k % 2))
        fl = np.array(c["flags"], dtype=bool)
        dnu = np.nanmax(np.abs(r["nu_refs"][0] / np.array(o["nu"], dtype=float) - 1.0))
        # only where both stopped at the same point do the remaining outputs compare tightly
        d = r["params"][0] - o["params"]; d[0] = (d[0] + 0.5) % 1.0 - 0.5
        same = np.all(np.abs(d)[fl] <= 1e-7 * np.maximum(o["errs"][fl], 1e-12) + 1e-11) and dnu < 1e-9
        if not same:
            skipped += 1
            # (parameters referred to different output frequencies differ trivially: is it the
            # iterate, or the zero-covariance frequency itself?)
            if dnu >= 1e-6:
                print("  case %d flags %s log10 %s option %d is_toa %d: nu_refs %s vs oracle %s" % (
                    k, "".join(map(str, c["flags"])), c["l10"], c["option"], k % 2, np.array2string(r["nu_refs"][0], precision=6),
                    np.array2string(np.array(o["nu"], dtype=float), precision=6)))
            continue
        ifit = np.where(fl)[0]
        cov_d = r["cov"][0][np.ix_(ifit, ifit)]
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / (np.abs(np.asarray(b)) + 1e-300)))
        vals = dict(nu_refs=dnu, param_errs=rel(r["param_errs"][0][fl], o["errs"][fl]), chi2=rel(r["chi2"][0], o["chi2"]),
                    red_chi2=rel(r["red_chi2"][0], o["red_chi2"]), snr=rel(r["snr"][0], o["snr"]),
                    scales=rel(r["scales"][0][ok], o["scales"]), scale_errs=rel(r["scale_errs"][0][ok], o["scale_errs"]),
                    channel_snrs=rel(r["channel_snrs"][0][ok], o["csnr"]),
                    covariance=float(np.max(np.abs(cov_d - o["cov"]) / np.sqrt(np.outer(np.diag(o["cov"]), np.diag(o["cov"]))))))
        for name, v in vals.items():
            w = worst.setdefault(name, (0.0, -1))
            if v > w[0]:
                worst[name] = (v, k)
    print("%d cases, %d skipped (the two stopped at different iterates)" % (n, skipped))
    for name, (v, k) in worst.items():
        c = make_case(k) if k >= 0 else None
        print("  %-13s worst relative difference %.2e  (case %d, flags %s, log10 %s, option %s)" % (
            name, v, k, "".join(map(str, c["flags"])) if c else "-", c["l10"] if c else "-", c["option"] if c else "-"))
