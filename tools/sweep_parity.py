"""A large randomised parity sweep (not part of the test suite): N small problems of random
shape, mask, noise, flag family (scattering included), tau parametrisation and option,
each fitted by the CPU oracle (a process pool, started before the GPU is touched) and by
the device with method 'trust-ncg' (raw parity) and 'newton' (optimum).  Prints the
distribution of the raw differences and the worst cases.  (GPU box)
    python tools/sweep_parity.py [ncases] [workers]"""
import multiprocessing as mp
import os, sys, time
import numpy as np
sys.path.insert(0, ".")

FLAGS = [([1, 1, 0, 0, 0], False), ([1, 0, 0, 0, 0], False), ([1, 1, 1, 0, 0], False), ([1, 0, 1, 0, 0], False),
         ([1, 1, 0, 1, 1], True), ([1, 1, 0, 1, 0], True), ([1, 0, 0, 1, 1], True), ([0, 0, 0, 1, 1], True),
         ([1, 1, 1, 1, 0], True), ([1, 1, 1, 1, 1], True)]


def make_case(k):
    from tests.synth_host import make_inputs, caller_guess, model_portrait
    rng = np.random.default_rng(90000 + k)
    flags, scat = FLAGS[k % len(FLAGS)]
    C = int(rng.integers(6, 48))
    lo, hi = (int(v) for v in os.environ.get("PP_SWEEP_LOG2NBIN", "6,11").split(","))   # e.g. "11,14": 2048..8192
    nbin = int(2 ** rng.integers(lo, hi))
    l10 = bool(rng.random() < 0.6) if scat else False
    tau_us = float(rng.uniform(15.0, 45.0)) if scat else None
    freqs, model = model_portrait(C, nbin)
    inp = make_inputs(C, nbin, 70000 + k, model=model, DM0=(34.56789 if rng.random() < 0.3 else 0.0),
                      sigma=float(rng.choice([0.03, 0.08, 0.2])), scint=bool(rng.random() < 0.4),
                      GM=(0.25 if flags[2] else None), tau_us=tau_us)
    g = caller_guess(inp, fit_scat=scat, log10_tau=l10,
                     tau_guess_rot=(float(rng.uniform(0.7, 1.5)) * tau_us * 1e-6 / inp["P"]) if scat else None)
    mask = (rng.random(C) > 0.12).astype(np.uint8)
    if mask.sum() < 4:
        mask[:4] = 1
    errs = inp["errs"] * rng.uniform(0.7, 1.4, C)
    return dict(k=k, flags=flags, l10=l10, C=C, nbin=nbin, data=inp["data"], model=model, freqs=freqs, P=inp["P"],
                x0=g["init_params"], nu_fit=g["nu_fit"], mask=mask, errs=errs, option=int(rng.integers(0, 2)))


def oracle_fit(k):
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    from oracle import pptoas_oracle as orc
    c = make_case(k)
    ok = np.where(c["mask"])[0]
    nus = [c["nu_fit"]] * 3
    o = orc.fit_portrait_full(c["data"][ok], c["model"][ok], c["x0"], c["P"], c["freqs"][ok], nus, nus, c["errs"][ok],
                              c["flags"], log10_tau=c["l10"], option=c["option"])
    return k, np.asarray(o.params), np.asarray(o.param_errs), o.chi2, o.nfeval, o.return_code


if __name__ == "__main__":
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:
        ora = {r[0]: r[1:] for r in pool.imap_unordered(oracle_fit, range(ncases), chunksize=4)}
    print("oracle: %d fits in %.1f s" % (ncases, time.time() - t0))
    from pulseportraiture_amd.engine import Engine
    eng = Engine(0)
    for kv in os.environ.get("PP_SWEEP_OPTS", "").split():      # e.g. PP_SWEEP_OPTS="taylor=0 scat_model=0"
        name, _, val = kv.partition("=")
        eng.set_option(name, float(val))
    rows = []
    for k in range(ncases):
        c = make_case(k)
        eng.set_model(c["model"])
        kw = dict(errs=c["errs"][None], chan_mask=c["mask"][None], nu_fits=[[c["nu_fit"]] * 3],
                  nu_outs=[[c["nu_fit"]] * 3], fit_flags=c["flags"], log10_tau=c["l10"], option=c["option"])
        rn = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], **kw)
        rw = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], method='newton', **kw)
        # A/B of how the one-pass flow mirrors SciPy's one-point cache when it counts nfeval (option nfev_shadow)
        eng.set_option("nfev_shadow", 1)
        rs = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], **kw)
        eng.set_option("nfev_shadow", 2)
        rs2 = eng.fit_batch(c["data"][None], c["freqs"], c["P"], c["x0"], **kw)
        eng.set_option("nfev_shadow", 0)
        d2 = rs2["params"][0] - ora[k][0]
        d2[0] = (d2[0] + 0.5) % 1.0 - 0.5
        op, oe, ochi2, onfev, orc_ = ora[k]
        d = rn["params"][0] - op
        d[0] = (d[0] + 0.5) % 1.0 - 0.5
        dw = rw["params"][0] - op
        dw[0] = (dw[0] + 0.5) % 1.0 - 0.5
        sig = np.where(oe > 0, oe, 1.0)
        rows.append(dict(k=k, flags="".join(map(str, c["flags"])), l10=c["l10"], C=c["C"], nbin=c["nbin"],
                         params=[float(v) for v in rn["params"][0]], params_newton=[float(v) for v in rw["params"][0]],
                         oparams=[float(v) for v in op],
                         dphi=abs(d[0]), dDM=abs(d[1]), dsig=np.max(np.abs(d) / sig),
                         dphi_newton=abs(dw[0]), nfev=int(rn["nfeval"][0]), onfev=int(onfev), nfev_shadow=int(rs["nfeval"][0]),
                         nfev_shadow2=int(rs2["nfeval"][0]), dphi_shadow2=abs(d2[0]),
                         rc=int(rn["return_code"][0]), orc=int(orc_), chi2rel=abs(rn["chi2"][0] / ochi2 - 1.0)))
    # per-case rows, the device's raw answers included: tools/ref_self_scatter.py (build container)
    # sets the TRUE reference's own reproducibility beside them
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.environ.get("PP_SWEEP_JSON", "gpurun_out/parity_sweep_rows.json"), "w") as fh:
        json.dump(dict(ncases=ncases, log2nbin=os.environ.get("PP_SWEEP_LOG2NBIN", "6,11"),
                       opts=os.environ.get("PP_SWEEP_OPTS", ""), rows=rows), fh, default=float)
    dphi = np.array([r["dphi"] for r in rows]); dsig = np.array([r["dsig"] for r in rows])
    print("trust-ncg raw |dphi|: median %.1e  90%% %.1e  99%% %.1e  max %.1e   (< 1e-10: %.1f %%, < 1e-9 [the bar]: %.1f %%)" % (
        np.median(dphi), np.percentile(dphi, 90), np.percentile(dphi, 99), dphi.max(), 100 * (dphi < 1e-10).mean(),
        100 * (dphi < 1e-9).mean()))
    ddm = np.array([r["dDM"] for r in rows])
    print("trust-ncg raw |dDM|: median %.1e  99%% %.1e  max %.1e   (< 1e-6 [the bar]: %.1f %%)" % (
        np.median(ddm), np.percentile(ddm, 99), ddm.max(), 100 * (ddm < 1e-6).mean()))
    print("max |dparam|/sigma: median %.1e  99%% %.1e  max %.1e" % (np.median(dsig), np.percentile(dsig, 99), dsig.max()))
    print("newton |dphi| vs the reference's raw answer: median %.1e  max %.1e" % (
        np.median([r["dphi_newton"] for r in rows]), max(r["dphi_newton"] for r in rows)))
    fam = {}
    for r in rows:
        fam.setdefault((r["flags"], r["l10"]), []).append(r)
    for key, rs in sorted(fam.items()):
        dp = np.array([r["dphi"] for r in rs])
        scat = key[0][3] == "1" or key[0][4] == "1"
        same = np.mean([r["nfev"] == r["onfev"] for r in rs])
        same_sh = np.mean([r["nfev_shadow"] == r["onfev"] for r in rs])
        same_sh2 = np.mean([r["nfev_shadow2"] == r["onfev"] for r in rs])
        dp2 = np.array([r["dphi_shadow2"] for r in rs])
        print("  %s log10=%d  n=%3d  |dphi| median %.1e max %.1e  >1e-10: %2d  nfeval = ref's: %.0f %% (nfev_shadow=1: %.0f %%, =2: %.0f %% with >1e-10: %2d)  rc!=2: %d" % (
            key[0], key[1], len(rs), np.median(dp), dp.max(), (dp >= 1e-10).sum(), 100 * same, 100 * same_sh, 100 * same_sh2,
            (dp2 >= 1e-10).sum(), sum(r["rc"] != 2 for r in rs)))
    worst = sorted(rows, key=lambda r: -r["dphi"])[:12]
    for r in worst:
        print("  worst: case %d %s l10=%d C=%d nbin=%d dphi %.2e dDM %.2e dsig %.2e nfev %d/%d rc %d/%d chi2rel %.1e" % (
            r["k"], r["flags"], r["l10"], r["C"], r["nbin"], r["dphi"], r["dDM"], r["dsig"], r["nfev"], r["onfev"],
            r["rc"], r["orc"], r["chi2rel"]))
