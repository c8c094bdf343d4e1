#!/usr/bin/env python3
"""Caller-level goldens for the get_TOAs options the first set (make_golden_gettoas.py)
does not reach -- run the TRUE reference's GetTOAs.get_TOAs with

    gettoas_opt_two_archives   a `datafiles` list of two archives (different DM, nsub):
                               per-archive DeltaDM means, the order of TOA_list
                               (pptoas.py:247, 665-721)
    gettoas_opt_DM0            DM0= given (pptoas.py:315-318, 665)
    gettoas_opt_fixalpha       fit_scat=True, fix_alpha=True (pptoas.py:216-227)
    gettoas_opt_lintau         fit_scat=True, log10_tau=False (pptoas.py:448-450, 614-627)
    gettoas_opt_nufits         nu_fits=(nu1, nu2) with a scattering fit (pptoas.py:402-407)

and store the inputs and EVERY result list of the GetTOAs object (all 43 of
pptoas.py:101-147, per archive) plus every TOA of TOA_list (archive, frequency, MJD,
error, DM, DM error, flags in insertion order) in tests/golden/gettoas_opt_*.npz.

Build-container only (needs /root/reference): see make_golden.py / make_golden_gettoas.py
for how the reference is converted and PSRCHIVE is patched away.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_gettoas as mgt  # noqa: E402

# the list attributes GetTOAs.__init__ creates (pptoas.py:101-147) that hold one numeric
# entry per archive
PER_ARCHIVE = ["doppler_fs", "nu0s", "nu_fits", "nu_refs", "ok_isubs", "MJDs", "Ps", "phis", "phi_errs",
               "TOA_errs", "DM0s", "DMs", "DM_errs", "DeltaDM_means", "DeltaDM_errs", "GMs", "GM_errs",
               "taus", "tau_errs", "alphas", "alpha_errs", "scales", "scale_errs", "snrs", "channel_snrs",
               "profile_fluxes", "profile_flux_errs", "fluxes", "flux_errs", "flux_freqs", "red_chi2s",
               "covariances", "nfevals", "rcs"]


def _num(v):
    if v is None:
        return None
    if isinstance(v, (str, bytes)):
        return str(v)
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    return float(v)


def dump(gt, names):
    out = {}
    narch = len(gt.order)
    out["narch"] = narch
    out["order"] = np.array([str(o) for o in gt.order])
    out["ok_idatafiles"] = np.asarray(gt.ok_idatafiles)
    out["nfit"] = gt.nfit
    out["fit_flags"] = np.asarray(gt.fit_flags)
    out["n_fit_durations"] = len(gt.fit_durations)
    for ia in range(narch):
        for name in PER_ARCHIVE:
            v = getattr(gt, name)[ia]
            if name in ("nu_fits", "nu_refs"):
                v = np.array([[np.nan if x is None else float(x) for x in row] for row in v])
            out["a%d_%s" % (ia, name)] = np.asarray(v, dtype=np.float64)
        out["a%d_TOA_days" % ia] = np.array([t.d if t != 0 else 0 for t in gt.TOAs[ia]])
        out["a%d_TOA_fracs" % ia] = np.array([t.f if t != 0 else 0.0 for t in gt.TOAs[ia]])
        out["a%d_epoch_days" % ia] = np.array([e.d for e in gt.epochs[ia]])
        out["a%d_epoch_fracs" % ia] = np.array([e.f for e in gt.epochs[ia]])
        o = gt.obs[ia]
        out["a%d_obs" % ia] = np.array([str(o.telescope), str(o.backend), str(o.frontend)])
    toas = []
    for t in gt.TOA_list:
        toas.append(dict(archive=str(t.archive), frequency=_num(t.frequency), day=int(t.MJD.d), frac=float(t.MJD.f),
                         TOA_error=_num(t.TOA_error), telescope=str(t.telescope),
                         telescope_code=str(t.telescope_code), DM=_num(t.DM), DM_error=_num(t.DM_error),
                         flags=[[str(k), _num(v)] for k, v in t.flags.items()]))
    out["TOA_list_json"] = np.array(json.dumps(toas))
    return out


def main():
    ref, pptoas, tmp = mgt.import_pptoas()
    model = os.path.join(mg.REF, "examples", "example.gmodel")
    cases = [
        ("gettoas_opt_two_archives",
         [dict(seed=41, nsub=3, DM0=34.56789), dict(seed=42, nsub=4, DM0=12.345678, sigma=0.08)],
         dict(print_phase=True)),
        ("gettoas_opt_DM0", [dict(seed=43, nsub=4)], dict(DM0=34.5, bary=False)),
        ("gettoas_opt_fixalpha", [dict(seed=44, nsub=4, tau_us=20.0)],
         dict(fit_scat=True, fix_alpha=True, scat_guess=(30e-6, 1500.0, -4.0))),
        ("gettoas_opt_lintau", [dict(seed=45, nsub=4, tau_us=20.0)],
         dict(fit_scat=True, log10_tau=False, scat_guess=(30e-6, 1500.0, -4.0))),
        ("gettoas_opt_nufits", [dict(seed=46, nsub=4, tau_us=20.0)],
         dict(fit_scat=True, log10_tau=True, scat_guess=(30e-6, 1500.0, -4.0), nu_fits=(1450.0, 1550.0),
              nu_refs=(1500.0, 1400.0))),
    ]
    # subints with one and with two usable channels under fit_GM (pptoas.py:475-486): one channel -> phase only; two
    # channels -> "fit_flags[2] = 0" on the list LEFT OVER from the previous subint -- [1,0,0,0,0] right after a
    # one-channel subint (so that two-channel subint is fitted for phase only), [1,1,1,0,0] -> [1,1,0,0,0] after a normal one
    cases.append(("gettoas_opt_fewchan", [dict(seed=47, nsub=6, GM=0.25, fewchan=True)], dict(fit_GM=True, bary=False)))
    # fit_DM=False (phase only everywhere: DM / DM_err None on the TOA lines, unit DM weights in the archive's mean) with the
    # parallactic-angle flag and caller-supplied flags (pptoas.py:196-215, 606-608, 646-651, 665-682)
    cases.append(("gettoas_opt_nodm", [dict(seed=48, nsub=4)],
                  dict(fit_DM=False, print_parangle=True, addtnl_toa_flags={"pta": "NANOGrav", "ver": 0.1})))
    # the flux estimate of a scattering fit (pptoas.py:554-575: the template scattered by the FITTED tau and alpha before its means are taken)
    cases.append(("gettoas_opt_scatflux", [dict(seed=49, nsub=4, tau_us=20.0)],
                  dict(fit_scat=True, print_flux=True, scat_guess=(30e-6, 1500.0, -4.0))))
    for name, archives, gkw in cases:
        bunches, store = {}, {}
        names = []
        for ia, skw in enumerate(archives):
            skw = dict(skw)
            nsub = skw.get("nsub", 5)
            fewchan = skw.pop("fewchan", False)
            data, arrays, scal = mgt.synth_archive(ref, **skw)
            if fewchan:
                w = np.ones_like(arrays["weights"])
                w[1, :] = 0.0; w[1, 10] = 1.0
                w[2, :] = 0.0; w[2, [5, 25]] = 1.0
                w[4, :] = 0.0; w[4, [3, 30]] = 1.0
                arrays["weights"] = w
                data.weights = w
                data.ok_ichans = [np.where(w[i] > 0)[0] for i in range(nsub)]
                data.ok_isubs = np.array([i for i in range(nsub) if len(data.ok_ichans[i])])
                data.masks = (w > 0)[:, None, :, None]
            # (synth_archive zaps subint 2 entirely: keep that for archives that have one)
            fname = "fake_%d.fits" % ia
            data.filename = fname
            bunches[fname] = data
            names.append(fname)
            for k, v in arrays.items():
                store["in%d_%s" % (ia, k)] = v
            for k, v in scal.items():
                store["in%d_scal_%s" % (ia, k)] = np.asarray(v)
            store["in%d_filename" % ia] = np.array(fname)
        pptoas.load_data = lambda f, *a, **k: bunches[f]
        gt = pptoas.GetTOAs(names[0], model, quiet=True)
        gt.datafiles = list(names)
        gt.get_TOAs(quiet=True, **gkw)
        out = dump(gt, names)
        kwargs = {("kwjson_" + k if isinstance(v, dict) else "kw_" + k): (np.array(json.dumps(v)) if isinstance(v, dict) else np.asarray(v))
                  for k, v in gkw.items()}
        mg.save(name, narchives=len(archives), **store, **{"out_" + k: v for k, v in out.items()}, **kwargs)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
