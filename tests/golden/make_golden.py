#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by running the TRUE reference.

Runs only in the build container (needs /root/reference).  The reference is
Python 2; it is converted to Python 3 in a scratch directory OUTSIDE the repo
(lib2to3 + the five integer-division / dtype fixes of SURVEY.md Appendix B and
a two-line stub for the absent `psrchive` module), imported from there, and
only its numerical inputs/outputs are written here as .npz files.  No
reference source is copied into the repository.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Versions are recorded inside every fixture (numpy / scipy of the generating
run): the reference itself pins neither (setup.py:5-14).
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import scipy

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20260101


def import_reference():
    tmp = tempfile.mkdtemp(prefix="pp_ref_py3_")
    names = ["pplib.py", "pptoaslib.py", "pptoas.py", "telescope_codes.py"]
    for name in names:
        shutil.copy(os.path.join(REF, name), tmp)
    subprocess.run([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w",
                    "-n"] + names, cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    fixes = [("nharm = nbin/2 + 1", "nharm = nbin//2 + 1"),
             ("dtype='complex_'", "dtype=complex"),
             ("ngauss = (len(params) - 2) / 3",
              "ngauss = (len(params) - 2) // 3"),
             ("ngauss = (len(model_params) - 2) / 6",
              "ngauss = (len(model_params) - 2) // 6"),
             ("nsin = len(params)/3", "nsin = len(params)//3")]
    for name in ("pplib.py", "pptoaslib.py"):
        path = os.path.join(tmp, name)
        src = open(path).read()
        for a, b in fixes:
            src = src.replace(a, b)
        open(path, "w").write(src)
    open(os.path.join(tmp, "psrchive.py"), "w").write(
        "class MJD(object):\n    pass\n")
    os.environ["MPLBACKEND"] = "Agg"
    sys.path.insert(0, tmp)
    import pptoaslib  # noqa: E402  (star-imports pplib)
    return pptoaslib, tmp


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def versions():
    return dict(numpy_version=np.__version__, scipy_version=scipy.__version__)


def make_inputs(ref, C, B, seed, DM0=0.0, sigma=0.05, scint=False, GM=None,
                tau_us=None, alpha=-4.0, nu0=1500.0, bw=800.0):
    """Synthetic subint the way SURVEY.md 8(d) describes, built with the
    reference's own functions (read_model -> gen_gaussian_portrait,
    rotate_portrait_full, scattering_portrait_FT)."""
    rng = np.random.default_rng(seed)
    P = 1.0 / 345.67890123456789
    d = bw / C
    freqs = np.linspace(nu0 - bw / 2 + d / 2, nu0 + bw / 2 - d / 2, C)
    phases = ref.get_bin_centers(B)
    name, ngauss, model = ref.read_model(
        os.path.join(REF, "examples", "example.gmodel"), phases, freqs, P,
        quiet=True)
    phi_inj = rng.uniform(-0.5, 0.5)
    dDM_inj = rng.normal(3e-4, 2e-4)
    GM_inj = 0.0 if GM is None else rng.normal(GM, 0.05)
    port = model.copy()
    if tau_us is not None:
        tau_rot = tau_us * 1e-6 / P
        taus = ref.scattering_times(tau_rot, alpha, freqs, nu0)
        port = np.fft.irfft(ref.scattering_portrait_FT(taus, B) *
                            np.fft.rfft(port, axis=-1), axis=-1)
    # data "delayed by (phi, DM)" = rotate by the negatives, referenced to inf
    port = ref.rotate_portrait_full(port, -phi_inj, -(DM0 + dDM_inj), -GM_inj,
                                    freqs, np.inf, np.inf, P)
    if scint:
        pars = []
        for _ in range(3):
            pars += [rng.uniform(0, 1.0), rng.chisquare(5.0),
                     rng.uniform(0, 1)]
        port = ref.add_scintillation(port, params=pars)
    data = port + rng.normal(0.0, sigma, size=port.shape)
    errs = np.full(C, sigma)
    return dict(data=data, model=model, freqs=freqs, errs=errs, P=P,
                phi_inj=phi_inj, dDM_inj=dDM_inj, DM0=DM0, GM_inj=GM_inj,
                nu0=nu0, bw=bw, sigma=sigma)


def caller_guess(ref, inp, fit_scat=False, log10_tau=True, tau_guess_rot=None,
                 alpha_guess=-4.0):
    """The get_TOAs preamble (pptoas.py:399-460) with SNRs = 1."""
    freqs, P = inp["freqs"], inp["P"]
    nu_mean = freqs.mean()
    nu_fit = ref.guess_fit_freq(freqs, None)
    DM_guess = inp["DM0"]
    rot_port = ref.rotate_data(inp["data"], 0.0, DM_guess, P, freqs, nu_mean)
    rot_prof = np.average(rot_port, axis=0, weights=np.ones(len(freqs)))
    B = inp["data"].shape[1]
    tau_guess, a_guess = 0.0, 0.0
    mprof = inp["model"].mean(axis=0)
    if fit_scat:
        a_guess = alpha_guess
        tau_guess = 0.0 if tau_guess_rot is None else tau_guess_rot
        mprof = np.fft.irfft(ref.scattering_portrait_FT(
            np.array([tau_guess]), B)[0] * np.fft.rfft(mprof))
    fps = ref.fit_phase_shift(rot_prof, mprof, Ns=100)
    if fit_scat and log10_tau:
        if tau_guess == 0.0:
            tau_guess = B ** -1
        tau_guess = np.log10(tau_guess)
    phi_guess = ref.phase_transform(fps.phase, DM_guess, nu_mean, nu_fit, P,
                                    mod=True)
    return dict(nu_fit=nu_fit, nu_mean=nu_mean, rot_prof=rot_prof,
                model_prof=mprof, fps_phase=fps.phase,
                fps_phase_err=fps.phase_err, fps_scale=fps.scale,
                fps_scale_err=fps.scale_err, fps_snr=fps.snr,
                fps_red_chi2=fps.red_chi2,
                init_params=np.array([phi_guess, DM_guess, 0.0, tau_guess,
                                      a_guess], dtype=np.float64))


def result_dict(r, prefix="out_"):
    out = {}
    for key in ("params", "param_errs", "phi", "phi_err", "DM", "DM_err",
                "GM", "GM_err", "tau", "tau_err", "alpha", "alpha_err",
                "scales", "scale_errs", "nu_DM", "nu_GM", "nu_tau",
                "covariance_matrix", "chi2", "red_chi2", "snr",
                "channel_snrs", "nfeval", "return_code"):
        out[prefix + key] = np.asarray(r[key], dtype=np.float64)
    return out


def objective_points(ref, inp, g, flags, log10_tau, nu_fits):
    """f, grad, Hessian of the reference objective at three fixed points."""
    B = inp["data"].shape[1]
    dFT = np.fft.rfft(inp["data"], axis=-1)
    dFT[:, 0] *= ref.F0_fact
    mFT = np.fft.rfft(inp["model"], axis=-1)
    mFT[:, 0] *= ref.F0_fact
    errs_FT = inp["errs"] * np.sqrt(B / 2.0)
    x0 = g["init_params"].copy()
    pts = [x0,
           x0 + np.array([1.3e-3, 2.1e-4, 0.02, 0.03, 0.11]),
           x0 + np.array([-4.0e-4, -1.0e-4, -0.01, -0.05, -0.2])]
    args = (dFT, mFT, errs_FT, inp["P"], inp["freqs"], nu_fits[0], nu_fits[1],
            nu_fits[2], np.array(flags, dtype=bool), log10_tau)
    f = np.array([ref.fit_portrait_full_function(p, *args) for p in pts])
    gr = np.array([ref.fit_portrait_full_function_deriv(p, *args)
                   for p in pts])
    hs = np.array([ref.fit_portrait_full_function_2deriv(p, *args)
                   for p in pts])
    Sd = ((np.abs(dFT) ** 2).T / errs_FT ** 2.0).T.sum()
    return dict(obj_points=np.array(pts), obj_f=f, obj_grad=gr, obj_hess=hs,
                obj_Sd=Sd)


def save(name, **arrays):
    arrays.update({k: np.asarray(v) for k, v in versions().items()})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f kB)" % (path, os.path.getsize(path) / 1e3))


def fit_case(ref, name, C, B, seed, flags, log10_tau=False, nu_outs=None,
             store_arrays=True, patch_cov=False, option=0, is_toa=True, **kw):
    fit_scat = bool(flags[3] or flags[4])
    inp = make_inputs(ref, C, B, seed, **kw)
    tau_rot = None
    if kw.get("tau_us") is not None:
        # pptoas seeds tau from the model file / scat_guess; use a deliberately
        # offset guess (x1.5) at nu_fit, alpha = -4
        tau_rot = 1.5 * kw["tau_us"] * 1e-6 / inp["P"]
    g = caller_guess(ref, inp, fit_scat=fit_scat, log10_tau=log10_tau,
                     tau_guess_rot=tau_rot)
    nu_fits = [g["nu_fit"]] * 3
    if nu_outs is None:
        nu_outs = [None, None, None]
    r = ref.fit_portrait_full(inp["data"], inp["model"], g["init_params"],
                              inp["P"], inp["freqs"], nu_fits, nu_outs,
                              inp["errs"], flags, [(None, None)] * 5,
                              log10_tau, option=option, sub_id=None,
                              method='trust-ncg', is_toa=is_toa, quiet=True)
    out = dict(C=C, B=B, seed=seed, fit_flags=np.array(flags), option=option,
               is_toa=is_toa,
               log10_tau=log10_tau, P=inp["P"], freqs=inp["freqs"],
               errs=inp["errs"], nu_fits=np.array(nu_fits, dtype=np.float64),
               nu_outs=np.array([np.nan if v is None else v for v in nu_outs]),
               inj=np.array([inp["phi_inj"], inp["DM0"] + inp["dDM_inj"],
                             inp["GM_inj"]]),
               input_sha=sha(inp["data"], inp["model"]),
               patched_covariance=patch_cov)
    out.update({k: v for k, v in g.items()})
    out.update(result_dict(r))
    if not store_arrays:
        # inputs are regenerated from the seed by tests/synth_host.py; these
        # summaries check the regeneration (bitwise equality is not attainable:
        # numpy's complex multiply rounds differently with buffer alignment)
        out.update(data_rowsum=inp["data"].sum(1), data_colsum=inp["data"].sum(0),
                   data_sample=inp["data"][::16, ::16],
                   model_sample=inp["model"][::16, ::16])
    if store_arrays:
        out.update(data=inp["data"], model=inp["model"])
        out.update(objective_points(ref, inp, g, flags, log10_tau, nu_fits))
    save(name, **out)
    return inp, g, r


def main():
    ref, tmp = import_reference()
    assert ref.Dconst == 0.000241 ** -1
    # ---- fit_portrait_full, 64x256 (cfg1 shape), every flag family ----
    fit_case(ref, "fpf_64x256_phiDM", 64, 256, SEED + 1, [1, 1, 0, 0, 0])
    fit_case(ref, "fpf_64x256_phiDM_dm0", 64, 256, SEED + 2, [1, 1, 0, 0, 0],
             DM0=34.56789)
    fit_case(ref, "fpf_64x256_phiDMGM", 64, 256, SEED + 3, [1, 1, 1, 0, 0],
             GM=0.25)
    fit_case(ref, "fpf_64x256_scat", 64, 256, SEED + 4, [1, 1, 0, 1, 1],
             log10_tau=True, tau_us=20.0)
    fit_case(ref, "fpf_64x256_phiDMtau", 64, 256, SEED + 5, [1, 1, 0, 1, 0],
             log10_tau=True, tau_us=20.0)
    fit_case(ref, "fpf_64x256_scat_lin", 64, 256, SEED + 6, [1, 1, 0, 1, 1],
             log10_tau=False, tau_us=30.0)
    fit_case(ref, "fpf_64x256_phi", 64, 256, SEED + 7, [1, 0, 0, 0, 0])
    fit_case(ref, "fpf_64x256_nuout", 64, 256, SEED + 8, [1, 1, 0, 0, 0],
             nu_outs=[1400.0, 1400.0, 1400.0])
    fit_case(ref, "fpf_64x256_lowsnr_scint", 64, 256, SEED + 9,
             [1, 1, 0, 0, 0], sigma=1.5, scint=True)
    fit_case(ref, "fpf_64x256_all5", 64, 256, SEED + 10, [1, 1, 1, 1, 1],
             log10_tau=True, tau_us=20.0, GM=0.25)
    fit_case(ref, "fpf_128x512_phiDM_scint", 128, 512, SEED + 11,
             [1, 1, 0, 0, 0], scint=True, DM0=34.56789)
    # ---- larger shape, scalar outputs only (inputs regenerated from seed by
    # the oracle's own generator; input_sha guards against drift) ----
    fit_case(ref, "fpf_512x1024_phiDM_scalars", 512, 1024, SEED + 12,
             [1, 1, 0, 0, 0], store_arrays=False)
    # ---- fit_phase_shift and legacy fit_portrait ----
    inp = make_inputs(ref, 64, 256, SEED + 20)
    g = caller_guess(ref, inp)
    prof = g["rot_prof"]
    mprof = g["model_prof"]
    rows = []
    for shift in (0.0, 0.123, -0.37):
        d = ref.rotate_data(prof, -shift)
        for noise in (None, 0.05 / 8.0):
            r = ref.fit_phase_shift(d, mprof, noise=noise, Ns=100)
            rows.append([shift, np.nan if noise is None else noise, r.phase,
                         r.phase_err, r.scale, r.scale_err, r.snr,
                         r.red_chi2])
    save("fit_phase_shift_256", prof=prof, model_prof=mprof,
         rows=np.array(rows),
         row_fields=np.array(["shift", "noise", "phase", "phase_err", "scale",
                              "scale_err", "snr", "red_chi2"]))
    r = ref.fit_portrait(inp["data"], inp["model"], g["init_params"][:2],
                         inp["P"], inp["freqs"], g["nu_fit"], None,
                         inp["errs"], quiet=True)
    save("legacy_fit_portrait_64x256", data=inp["data"], model=inp["model"],
         freqs=inp["freqs"], errs=inp["errs"], P=inp["P"],
         init_params=g["init_params"][:2], nu_fit=g["nu_fit"],
         out_phase=r.phase, out_phase_err=r.phase_err, out_DM=r.DM,
         out_DM_err=r.DM_err, out_scales=r.scales,
         out_scale_errs=r.scale_errs, out_nu_ref=r.nu_ref,
         out_covariance=r.covariance, out_chi2=r.chi2,
         out_red_chi2=r.red_chi2, out_snr=r.snr)
    # ---- helper restatement checks: model portrait pieces ----
    P = 1.0 / 345.67890123456789
    freqs = np.linspace(1106.25, 1893.75, 64)
    phases = ref.get_bin_centers(256)
    _, _, model = ref.read_model(os.path.join(REF, "examples",
                                              "example.gmodel"), phases,
                                 freqs, P, quiet=True)
    save("helpers_64x256", freqs=freqs, phases=phases, model=model,
         gp=ref.gaussian_profile(256, 0.9961, 0.031),
         gp2=ref.gaussian_profile(256, 1.23, 0.11),
         nu_fit=ref.guess_fit_freq(freqs, None),
         nu_fit_snr=ref.guess_fit_freq(freqs, np.linspace(1.0, 3.0, 64)),
         noise_ps=ref.get_noise_PS(inp["data"], chans=True),
         phase_tr=ref.phase_transform(0.3, 34.5, 1500.0, 1200.0, P, mod=True))
    # ---- spline (PCA + B-spline) model portraits: gen_spline_portrait ----
    import pickle
    import scipy.interpolate as si
    rng = np.random.default_rng(SEED + 40)
    nb, ncomp = 256, 2
    mean_prof = model.mean(axis=0)
    u, sv, vt = np.linalg.svd(model - mean_prof, full_matrices=False)
    eigvec = vt[:ncomp].T                                   # [nbin, ncomp]
    proj = np.dot(model - mean_prof, eigvec)                # [nchan, ncomp]
    (tck, uu), fp, ier, msg = si.splprep(proj.T, u=freqs, k=3, s=len(freqs) * 1e-6,
                                         full_output=True)
    spl_path = os.path.join(HERE, "example.spl")
    with open(spl_path, "wb") as fh:
        pickle.dump(["example_spline", "J1234-5678", "fake.fits", mean_prof, eigvec, tck],
                    fh, protocol=2)
    f_eval = np.linspace(1110.0, 1890.0, 40)
    save("spline_model_256", freqs=f_eval,
         port=ref.gen_spline_portrait(mean_prof, f_eval, eigvec, tck, None),
         port_512=ref.gen_spline_portrait(mean_prof, f_eval, eigvec, tck, 512),
         port_flat=ref.gen_spline_portrait(mean_prof, f_eval, eigvec[:, :0], tck, None))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
