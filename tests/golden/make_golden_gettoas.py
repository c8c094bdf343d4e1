#!/usr/bin/env python3
"""Caller-level goldens: run the TRUE reference's GetTOAs.get_TOAs on a synthetic
archive (DataBunch built from arrays, PSRCHIVE monkey-patched away) and store
its inputs and per-subint outputs in tests/golden/gettoas_*.npz.

Build-container only (needs /root/reference); see make_golden.py for how the
reference is converted to Python 3 in a scratch directory.  Additional patch
here: the py2 `exec`-into-locals unpacking of the load_data DataBunch
(pptoas.py:278-279) is replaced by explicit assignments.
"""
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

KEYS = ["arch", "backend", "backend_delay", "bw", "doppler_factors", "DM", "dmc",
        "epochs", "filename", "flux_prof", "freqs", "frontend", "integration_length",
        "masks", "nbin", "nchan", "noise_stds", "npol", "nsub", "nu0", "ok_ichans",
        "ok_isubs", "parallactic_angles", "phases", "prof", "prof_noise", "prof_SNR",
        "Ps", "SNRs", "source", "state", "subints", "subtimes", "telescope",
        "telescope_code", "weights"]


class FakeMJD(object):
    def __init__(self, day=0, frac=0.0):
        if not isinstance(day, (int, np.integer)):
            w = np.floor(day)
            day, frac = int(w), float(day - w) + frac
        c = np.floor(frac)
        self.d, self.f = int(day) + int(c), float(frac - c)

    def in_days(self):
        return self.d + self.f

    def intday(self):
        return self.d

    def fracday(self):
        return self.f

    def __add__(self, o):
        return FakeMJD(self.d + o.d, self.f + o.f)


def import_pptoas():
    ref, tmp = mg.import_reference()
    path = os.path.join(tmp, "pptoas.py")
    src = open(path).read()
    pat = re.compile(r'^(\s*)for key in list\(data\.keys\(\)\):\n\s*exec\(key \+ " = data\[\'" \+ key \+ "\'\]"\)\n',
                     re.M)

    def repl(m):
        ind = m.group(1)
        return "".join("%s%s = data['%s']\n" % (ind, k, k) for k in KEYS)
    src, n = pat.subn(repl, src)
    assert n == 2, n
    open(path, "w").write(src)
    import psrchive
    psrchive.MJD = FakeMJD
    import pptoas
    pptoas.file_is_type = lambda f, t: False
    pptoas.pr.MJD = FakeMJD
    return ref, pptoas, tmp


def synth_archive(ref, seed, nsub=5, C=32, B=256, DM0=34.56789, sigma=0.05, GM=None,
                  tau_us=None, corrupt=False):
    rng = np.random.default_rng(seed)
    P0 = 1.0 / 345.67890123456789
    d = 800.0 / C
    freqs0 = np.linspace(1100 + d / 2, 1900 - d / 2, C)
    phases = ref.get_bin_centers(B)
    subints = np.zeros((nsub, 1, C, B))
    freqs = np.tile(freqs0, (nsub, 1))
    weights = np.ones((nsub, C))
    Ps = P0 * (1 + 1e-7 * np.arange(nsub))
    inj = []
    for i in range(nsub):
        _, _, model = ref.read_model(os.path.join(mg.REF, "examples", "example.gmodel"),
                                     phases, freqs[i], Ps[i], quiet=True)
        phi, dDM = rng.uniform(-0.5, 0.5), rng.normal(3e-4, 2e-4)
        gm = 0.0 if GM is None else rng.normal(GM, 0.05)
        port = model.copy()
        if tau_us is not None:
            taus = ref.scattering_times(tau_us * 1e-6 / Ps[i], -4.0, freqs[i], 1500.0)
            port = np.fft.irfft(ref.scattering_portrait_FT(taus, B) *
                                np.fft.rfft(port, axis=-1), axis=-1)
        port = ref.rotate_portrait_full(port, -phi, -(DM0 + dDM), -gm, freqs[i],
                                        np.inf, np.inf, Ps[i])
        subints[i, 0] = port + rng.normal(0, sigma, size=port.shape)
        if corrupt:
            # channels for get_channels_to_zap to find: narrow-band interference
            # (bad chi^2) and nearly dead channels (low S/N)
            subints[i, 0, 5 + i] += 0.3 * np.sin(2 * np.pi * 7 * phases + i)
            subints[i, 0, 20] = 0.01 * port[20] + rng.normal(0, sigma, size=B)
            if i == 3:
                subints[i, 0, 11] = 0.02 * port[11] + rng.normal(0, sigma, size=B)
        inj.append([phi, DM0 + dDM, gm])
        if i % 2:
            weights[i, rng.choice(C, size=4, replace=False)] = 0.0
    weights[2, :] = 0.0          # a fully zapped subint
    ok_ichans = [np.where(weights[i] > 0)[0] for i in range(nsub)]
    ok_isubs = np.array([i for i in range(nsub) if len(ok_ichans[i])])
    SNRs = rng.uniform(5, 50, size=(nsub, 1, C))
    epochs = [FakeMJD(55000 + i, 0.123456789012345 + 1e-3 * i) for i in range(nsub)]
    arrays = dict(subints=subints, freqs=freqs, weights=weights,
                  noise_stds=np.full((nsub, 1, C), sigma), SNRs=SNRs, Ps=Ps,
                  epoch_days=np.array([e.d for e in epochs]),
                  epoch_fracs=np.array([e.f for e in epochs]),
                  doppler_factors=1.0 + 1e-4 * rng.standard_normal(nsub),
                  subtimes=np.full(nsub, 60.0), parallactic_angles=np.zeros(nsub),
                  inj=np.array(inj))
    scal = dict(DM=DM0, dmc=0, backend_delay=1.25e-6, telescope="GBT", telescope_code="1",
                backend="GUPPI", frontend="Rcvr1_2", bw=800.0, nu0=1500.0,
                source="J1234-5678")
    data = ref.DataBunch(arch=None, filename="fake.fits", flux_prof=None,
                         integration_length=60.0 * nsub,
                         masks=(weights > 0)[:, None, :, None], nbin=B, nchan=C, npol=1,
                         nsub=nsub, ok_ichans=ok_ichans, ok_isubs=ok_isubs, phases=phases,
                         prof=None, prof_noise=None, prof_SNR=None, state="Intensity",
                         epochs=epochs, **{k: v for k, v in arrays.items()
                                           if k not in ("epoch_days", "epoch_fracs", "inj")},
                         **scal)
    return data, arrays, scal


def run(pptoas, data, ird=None, **kw):
    pptoas.load_data = lambda *a, **k: data
    gt = pptoas.GetTOAs("fake.fits", os.path.join(mg.REF, "examples", "example.gmodel"),
                        quiet=True)
    if ird is not None:
        gt.instrumental_response_dict = gt.ird = ird
    gt.get_TOAs(quiet=True, **kw)
    out = {}
    for name in ("phis", "phi_errs", "DMs", "DM_errs", "GMs", "GM_errs", "taus",
                 "tau_errs", "alphas", "alpha_errs", "snrs", "red_chi2s", "scales",
                 "scale_errs", "channel_snrs"):
        out[name] = np.asarray(getattr(gt, name)[0], dtype=np.float64)
    for name in ("profile_fluxes", "profile_flux_errs", "fluxes", "flux_errs", "flux_freqs"):
        out[name] = np.asarray(getattr(gt, name)[0], dtype=np.float64)
    # evaluation counts and return codes of every subint (pptoas.py:591-592, 719-720)
    out["nfevals"] = np.asarray(gt.nfevals[0], dtype=np.int64)
    out["rcs"] = np.asarray(gt.rcs[0], dtype=np.int64)
    out["nu_refs"] = np.array([list(map(float, r)) for r in gt.nu_refs[0]])
    out["nu_fits"] = np.array([list(map(float, r)) for r in gt.nu_fits[0]])
    out["ok_isubs"] = np.asarray(gt.ok_isubs[0])
    out["TOA_days"] = np.array([t.d if t != 0 else 0 for t in gt.TOAs[0]])
    out["TOA_fracs"] = np.array([t.f if t != 0 else 0.0 for t in gt.TOAs[0]])
    out["TOA_errs"] = np.asarray(gt.TOA_errs[0], dtype=np.float64)
    out["DeltaDM_mean"] = gt.DeltaDM_means[0]
    out["DeltaDM_err"] = gt.DeltaDM_errs[0]
    out["covariances"] = np.asarray(gt.covariances[0])
    t0 = gt.TOA_list[0]
    out["toa0_frequency"] = t0.frequency
    out["toa0_flag_names"] = np.array(sorted(t0.flags.keys()))
    out["toa0_flag_values"] = np.array([str(t0.flags[k]) for k in sorted(t0.flags.keys())])
    # per-channel goodness of fit and the channels it would zap (pptoas.py:1208-1285)
    gt.get_channels_to_zap(SNR_threshold=8.0, rchi2_threshold=1.3, iterate=True, show=False)
    nok, C = len(gt.ok_isubs[0]), data.nchan
    rc2 = np.full((nok, C), np.nan)
    zap = np.zeros((nok, C), dtype=bool)
    for j, isub in enumerate(gt.ok_isubs[0]):
        rc2[j, data.ok_ichans[isub]] = gt.channel_red_chi2s[0][j]
        zap[j, np.asarray(gt.zap_channels[0][j], dtype=int)] = True
    out["channel_red_chi2s"] = rc2
    out["zap_channels"] = zap
    return out


def main():
    ref, pptoas, tmp = import_pptoas()
    cases = [("gettoas_phiDM", dict(seed=31), dict()),
             ("gettoas_phiDM_nurefs", dict(seed=32), dict(nu_refs=(1400.0, 1400.0),
                                                        bary=False, print_phase=True)),
             ("gettoas_GM", dict(seed=33, GM=0.25), dict(fit_GM=True)),
             ("gettoas_scat", dict(seed=34, tau_us=20.0),
              dict(fit_scat=True, log10_tau=True, scat_guess=(30e-6, 1500.0, -4.0))),
             ("gettoas_zap", dict(seed=35, corrupt=True), dict(print_flux=True))]
    cases.append(("gettoas_ird", dict(seed=36), dict(add_instrumental_response=True)))
    for name, skw, gkw in cases:
        data, arrays, scal = synth_archive(ref, **skw)
        ird = None
        if name == "gettoas_ird":
            ird = {'DM': 34.56789, 'wids': [0.004, 0.003], 'irf_types': ['rect', 'gauss']}
        out = run(pptoas, data, ird=ird, **gkw)
        if ird is not None:
            import pptoaslib as ptl          # the converted reference module
            out["ird_resp"] = ptl.instrumental_response_port_FT(
                data.nbin, data.freqs[0][data.ok_ichans[0]], ird['DM'], data.Ps[0], ird['wids'],
                ird['irf_types'])
            out["ird_gauss_FT"] = ptl.gaussian_profile_FT(data.nbin, 0.3, 0.02, 1.7)
            out["ird_DM"] = ird['DM']
            out["ird_wids"] = np.array(ird['wids'])
            out["ird_types"] = np.array(ird['irf_types'])
        kwargs = {}
        for k, v in gkw.items():
            kwargs["kw_" + k] = np.asarray(v)
        mg.save(name, **arrays, **{"scal_" + k: np.asarray(v) for k, v in scal.items()},
                **{"out_" + k: v for k, v in out.items()}, **kwargs)
    # ---- narrowband TOAs (one per channel), same archive as gettoas_phiDM ----
    data, arrays, scal = synth_archive(ref, seed=31)
    pptoas.load_data = lambda *a, **k: data
    gt = pptoas.GetTOAs("fake.fits", os.path.join(mg.REF, "examples", "example.gmodel"), quiet=True)
    gt.get_narrowband_TOAs(quiet=True)
    out = {}
    for name in ("phis", "phi_errs", "scales", "scale_errs", "channel_snrs", "channel_red_chi2s",
                 "TOA_errs"):
        out[name] = np.asarray(getattr(gt, name)[0], dtype=np.float64)
    out["ok_isubs"] = np.asarray(gt.ok_isubs[0])
    out["TOA_days"] = np.array([[t.d if t != 0 else 0 for t in row] for row in gt.TOAs[0]])
    out["TOA_fracs"] = np.array([[t.f if t != 0 else 0.0 for t in row] for row in gt.TOAs[0]])
    out["ntoa"] = len(gt.TOA_list)
    t0 = gt.TOA_list[0]
    out["toa0_frequency"] = t0.frequency
    out["toa0_flag_names"] = np.array(sorted(t0.flags.keys()))
    mg.save("gettoas_narrowband", **arrays, **{"scal_" + k: np.asarray(v) for k, v in scal.items()},
            **{"out_" + k: v for k, v in out.items()})
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
