"""Reference-shaped wideband TOA driver: `GetTOAs` / `TOA` (pptoas.py:31-743).

`GetTOAs.get_TOAs` keeps the reference's arguments and result attributes but
replaces the per-subint Python loop (pptoas.py:344-489) by ONE batched device
call per archive: all good subints go to the engine as a [nsub, nchan, nbin]
batch with a channel mask (the per-subint ok_ichans), the phase seed
(pptoas.py:421-457) and the fit run in HIP kernels, and the TOA bookkeeping
(pptoas.py:528-721) is done here on the host.

PSRFITS I/O needs PSRCHIVE, which is outside this package: `datafiles` are
`DataBunch` objects with the fields of the reference's `load_data`
(pplib.py:2803-2813) -- see `data_from_arrays` -- or `.npz` files holding them.
"""
import sys
import time

import numpy as np

from . import gmodel
from .engine import EngineNotSupported, default_engine
from .pplib import DataBunch, guess_fit_freq, scattering_alpha, weighted_mean

max_nfile = 999
rm_baseline = True


# ---------------------------------------------------------------------------
# epochs: integer day + fraction of a day (PSRCHIVE's MJD keeps day, seconds
# and fractional seconds; a float64 fraction resolves 1e-16 d = 9 ps)
# ---------------------------------------------------------------------------
class MJD(object):
    def __init__(self, day=0, frac=0.0):
        if isinstance(day, MJD):
            day, frac = day._day, day._frac
        elif not isinstance(day, (int, np.integer)):
            whole = np.floor(day)
            day, frac = int(whole), float(day - whole) + float(frac)
        carry = np.floor(frac)
        self._day = int(day) + int(carry)
        self._frac = float(frac - carry)

    def intday(self):
        return self._day

    def fracday(self):
        return self._frac

    def in_days(self):
        return self._day + self._frac

    def __add__(self, other):
        other = other if isinstance(other, MJD) else MJD(other)
        return MJD(self._day + other._day, self._frac + other._frac)

    __radd__ = __add__

    def __sub__(self, other):
        other = other if isinstance(other, MJD) else MJD(other)
        return MJD(self._day - other._day, self._frac - other._frac)

    def __repr__(self):
        return "MJD(%d%s)" % (self._day, ("%.15f" % self._frac)[1:])


class TOA(object):
    """TOA attributes bundled together (pptoas.py:31-73)."""

    def __init__(self, archive, frequency, MJD, TOA_error, telescope,
                 telescope_code, DM=None, DM_error=None, flags={}):
        self.archive = archive
        self.frequency = frequency
        self.MJD = MJD
        self.TOA_error = TOA_error
        self.telescope = telescope
        self.telescope_code = telescope_code
        self.DM = DM
        self.DM_error = DM_error
        self.flags = flags
        for flag in flags.keys():
            setattr(self, flag, flags[flag])

    def write_TOA(self, inf_is_zero=True, outfile=None):
        write_TOAs(self, inf_is_zero=inf_is_zero, outfile=outfile, append=True)


def toa_string(toa, inf_is_zero=True):
    """One loosely IPTA-formatted TOA line (pplib.py:3465-3497)."""
    freq = 0.0 if (toa.frequency == np.inf and inf_is_zero) else toa.frequency
    s = "%s %.8f %d" % (toa.archive, freq, toa.MJD.intday()) + \
        ("%.15f   %.3f  %s" % (toa.MJD.fracday(), toa.TOA_error,
                               toa.telescope_code))[1:]
    if toa.DM is not None:
        s += " -pp_dm %.7f" % toa.DM
    if toa.DM_error is not None:
        s += " -pp_dme %.7f" % toa.DM_error
    for flag, value in toa.flags.items():
        if value is None:
            continue
        if hasattr(value, "lower"):
            s += " -%s %s" % (flag, value)
        elif 'int' in str(type(value)):
            s += " -%s %d" % (flag, value)
        elif flag.find("_cov") >= 0:
            s += " -%s %.1e" % (flag, value)
        elif flag.find("phs") >= 0:
            s += " -%s %.8f" % (flag, value)
        elif flag.find("flux") >= 0:
            s += " -%s %.5f" % (flag, value)
        else:
            s += " -%s %.3f" % (flag, value)
    return s


def write_TOAs(TOAs, inf_is_zero=True, SNR_cutoff=0.0, outfile=None, append=True):
    """Write TOAs to file or stdout (pplib.py:3445-3503); TOAs below the S/N
    cutoff (or without an snr flag) are skipped like filter_TOAs does."""
    toas = TOAs if hasattr(TOAs, "__len__") else [TOAs]
    toas = [t for t in toas if hasattr(t, "snr") and t.snr >= SNR_cutoff]
    lines = [toa_string(t, inf_is_zero) for t in toas]
    if outfile is not None:
        with open(outfile, 'a' if append else 'w') as of:
            for line in lines:
                of.write(line + "\n")
    else:
        for line in lines:
            print(line)


def data_from_arrays(subints, freqs, Ps, epochs, weights=None, noise_stds=None,
                     SNRs=None, DM=0.0, dmc=0, doppler_factors=None,
                     backend_delay=0.0, telescope="GBT", telescope_code="1",
                     backend="GUPPI", frontend="Rcvr", bw=None, nu0=None,
                     subtimes=None, parallactic_angles=None, source="noname",
                     filename="arrays"):
    """Build the DataBunch `load_data` would return (pplib.py:2650-2814) from
    arrays: subints[nsub,npol,nchan,nbin], freqs[nsub,nchan], Ps[nsub], epochs
    (MJD objects or floats).  Zero-weight channels are excluded like the
    reference does (pplib.py:2755-2757); noise defaults to the power-spectrum
    estimate per channel (get_noise_PS), computed on the device."""
    subints = np.asarray(subints)
    if subints.ndim == 3:
        subints = subints[:, None]
    nsub, npol, nchan, nbin = subints.shape
    freqs = np.broadcast_to(np.asarray(freqs, dtype=np.float64), (nsub, nchan)).copy()
    weights = np.ones((nsub, nchan)) if weights is None else \
        np.asarray(weights, dtype=np.float64)
    ok_ichans = [np.where(weights[i] > 0)[0] for i in range(nsub)]
    ok_isubs = np.array([i for i in range(nsub) if len(ok_ichans[i])], dtype=int)
    if SNRs is None:
        SNRs = np.ones((nsub, npol, nchan))
    epochs = [e if isinstance(e, MJD) else MJD(e) for e in epochs]
    Ps = np.asarray(Ps, dtype=np.float64)
    if doppler_factors is None:
        doppler_factors = np.ones(nsub)
    if subtimes is None:
        subtimes = np.zeros(nsub)
    if parallactic_angles is None:
        parallactic_angles = np.zeros(nsub)
    if bw is None:
        bw = (freqs[0].max() - freqs[0].min()) * nchan / max(nchan - 1, 1)
    if nu0 is None:
        nu0 = freqs[0].mean()
    from .pplib import get_bin_centers
    return DataBunch(
        subints=subints, freqs=freqs, weights=weights, noise_stds=noise_stds,
        SNRs=np.asarray(SNRs, dtype=np.float64), Ps=Ps, epochs=epochs,
        ok_isubs=ok_isubs, ok_ichans=ok_ichans, DM=DM, dmc=dmc,
        doppler_factors=np.asarray(doppler_factors, dtype=np.float64), nbin=nbin,
        nchan=nchan, nsub=nsub, npol=npol, phases=get_bin_centers(nbin),
        backend_delay=backend_delay, telescope=telescope,
        telescope_code=telescope_code, backend=backend, frontend=frontend, bw=bw,
        nu0=nu0, subtimes=np.asarray(subtimes, dtype=np.float64),
        integration_length=float(np.sum(subtimes)),
        parallactic_angles=np.asarray(parallactic_angles, dtype=np.float64),
        source=source, filename=filename, masks=(weights > 0)[:, None, :, None])


def _load(datafile):
    if isinstance(datafile, dict):
        return datafile, datafile.get("filename", "arrays")
    if str(datafile).endswith(".npz"):
        z = np.load(datafile, allow_pickle=True)
        kw = {k: z[k] for k in z.files}
        for k in ("DM", "dmc", "backend_delay", "telescope", "telescope_code",
                  "backend", "frontend", "source"):
            if k in kw:
                kw[k] = kw[k].item()
        kw.setdefault("filename", str(datafile))
        return data_from_arrays(**kw), str(datafile)
    raise RuntimeError("Cannot load_data(%s): PSRFITS archives need PSRCHIVE; pass a "
                       "DataBunch (data_from_arrays) or an .npz of its fields." % datafile)


def _dededisperse(eng, port, d, ok_isubs):
    """A DataBunch flagged dmc (stored dedispersed) is put back to its dispersed
    state before fitting, as the reference does by re-loading the archive with
    dededisperse=True (pptoas.py:256-265): every channel is delayed again by the
    stored DM relative to the centre frequency (what PSRCHIVE's dedisperse() took
    out).  Runs on the device (rotate_portraits)."""
    if not d.dmc:
        return port
    isubs = np.asarray(ok_isubs, dtype=int)
    return eng.rotate_portraits(port, d.freqs[isubs], d.Ps[isubs], DM=-float(d.DM),
                                nu_DM=float(d.nu0))


def _to_device_once(eng, port):
    """Host portraits as a device tensor when they fit comfortably in free HBM (so that
    two engine calls do not both pay the H->D copy); unchanged otherwise."""
    try:
        import torch
    except ImportError:
        return port
    if torch.is_tensor(port) or not torch.cuda.is_available():
        return port
    a = np.asarray(port)
    try:
        free_b, _ = torch.cuda.mem_get_info(eng.device)
        if a.nbytes * 2.5 > free_b:
            return port
        return torch.as_tensor(np.ascontiguousarray(a), device="cuda:%d" % eng.device)
    except RuntimeError:
        return port


def _noise_rows(d, isubs):
    """noise_stds[isubs, 0] of a DataBunch, measured from the power spectrum
    (pplib.get_noise, what load_data stores: pplib.py:2727-2731) when the bunch was
    built without them."""
    isubs = np.asarray(isubs, dtype=int)
    if d.noise_stds is not None:
        return np.ascontiguousarray(np.asarray(d.noise_stds)[isubs, 0], dtype=np.float64)
    from .pplib import get_noise
    sub = np.asarray(d.subints)
    return np.array([get_noise(sub[i, 0], chans=True) for i in isubs], dtype=np.float64)


def _take_subints(subints, ok_isubs):
    """[nok, nchan, nbin] total-intensity portraits of the good subints, without
    copying the archive when they are a contiguous run (fancy indexing would copy
    gigabytes before the H2D transfer even starts)."""
    a = np.asarray(subints)
    idx = np.asarray(ok_isubs, dtype=int)
    if len(idx) and np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx))):
        return np.ascontiguousarray(a[idx[0]:idx[0] + len(idx), 0])
    return np.ascontiguousarray(a[idx, 0])


class GetTOAs(object):
    """Measure wideband TOAs and DMs (pptoas.py:75-1419, the get_TOAs path)."""

    def __init__(self, datafiles, modelfile, quiet=False):
        if isinstance(datafiles, (list, tuple)):
            self.datafiles = list(datafiles)
        elif isinstance(datafiles, str) and not datafiles.endswith(".npz"):
            self.datafiles = [line.rstrip("\n") for line in open(datafiles)]
        else:
            self.datafiles = [datafiles]
        if len(self.datafiles) > max_nfile:
            print("Too many archives.  See/change max_nfile(=%d) in pptoas.py." % max_nfile)
            sys.exit()
        self.is_FITS_model = False
        self.modelfile = modelfile
        for name in ("obs", "doppler_fs", "nu0s", "nu_fits", "nu_refs", "ok_idatafiles",
                     "ok_isubs", "epochs", "MJDs", "Ps", "phis", "phi_errs", "TOAs",
                     "TOA_errs", "DM0s", "DMs", "DM_errs", "DeltaDM_means",
                     "DeltaDM_errs", "GMs", "GM_errs", "taus", "tau_errs", "alphas",
                     "alpha_errs", "scales", "scale_errs", "snrs", "channel_snrs",
                     "profile_fluxes", "profile_flux_errs", "fluxes", "flux_errs",
                     "flux_freqs", "red_chi2s", "channel_red_chi2s", "covariances",
                     "nfevals", "rcs", "fit_durations", "order", "TOA_list",
                     "zap_channels"):
            setattr(self, name, [])
        self.instrumental_response_dict = self.ird = \
            {'DM': 0.0, 'wids': [], 'irf_types': []}
        self.quiet = quiet

    # -- template ----------------------------------------------------------
    def _gmodel(self):
        """The parsed .gmodel, or None if the model file is not one (then it is
        tried as a spline model: the reference's fallback order, pptoas.py:352-379)."""
        mdl = self.modelfile if isinstance(self.modelfile, dict) else None
        if mdl is None:
            try:
                mdl = gmodel.read_gmodel(self.modelfile)
                if mdl["nu_ref"] is None or not mdl["ngauss"]:
                    mdl = None
            except (UnicodeDecodeError, ValueError, IndexError):
                mdl = None
        if mdl is not None:
            self.model_name, self.ngauss = mdl["name"], mdl["ngauss"]
            self.model_code, self.model_nu_ref = mdl["code"], mdl["nu_ref"]
            self.gparams, self.alpha = mdl["params"], mdl["alpha"]
        return mdl

    def _model_for(self, freqs_row, nbin, P, unscattered=False):
        """Template portrait (host array) at this subint's frequencies."""
        mdl = self._gmodel()
        if mdl is None:
            from .splmodel import read_spline_model
            self.model_name, port = read_spline_model(self.modelfile, freqs_row, nbin,
                                                      quiet=True)
            return port
        if unscattered:
            mdl = dict(mdl)
            mdl["params"] = mdl["params"].copy()
            mdl["params"][1] = 0.0
        return gmodel.gaussian_portrait(mdl, freqs_row, nbin, P)

    def _load_template(self, eng, slot, freqs_row, nbin, P, unscattered=False):
        """Put the template for these frequencies into a model slot of the engine:
        Gaussian-component models are synthesised on the device, spline models are
        evaluated on the host and uploaded."""
        mdl = self._gmodel()
        if mdl is None:
            # spline (.spl) template: B-spline curve x eigenvectors evaluated on the
            # device straight into the slot (no host portrait)
            from .splmodel import read_spline_model
            name, _, _, mean_prof, eigvec, tck = read_spline_model(self.modelfile, quiet=True)
            self.model_name = name
            if np.asarray(eigvec).reshape(len(mean_prof), -1).shape[1] <= 32 and \
                    (not np.asarray(eigvec).size or int(tck[2]) <= 5):
                eng.set_model_spline(mean_prof, eigvec, tck, freqs_row, nbin, slot=slot)
                return
        if mdl is None or mdl["ngauss"] > 64:
            eng.set_model(self._model_for(freqs_row, nbin, P, unscattered), slot=slot)
            return
        if unscattered:
            mdl = dict(mdl)
            mdl["params"] = mdl["params"].copy()
            mdl["params"][1] = 0.0
        eng.set_model_gaussian(mdl, freqs_row, nbin, P, slot=slot)

    def _reference_seed_inputs(self, port, d, ok_isubs, mask, tau_lin, nu_fit_tau, fit_scat, use_ird):
        """What the reference's phase guess is formed from (pptoas.py:421-457), per good
        subint: the weights of the channel mean, the mean frequency of the good channels and
        the template's mean profile over them (scattered by the guessed tau when fitting
        scattering): (w, nu_mean, mprofs)."""
        isubs = np.asarray(ok_isubs, dtype=int)
        nok, nchan, nbin = port.shape
        w = np.asarray(d.weights, dtype=np.float64)[isubs] * mask
        freqs = d.freqs[isubs]
        nu_mean = np.array([freqs[j, mask[j] > 0].mean() for j in range(nok)])
        mprofs, cache = np.empty((nok, nbin)), {}
        for j, isub in enumerate(isubs):
            key = (d.freqs[isub].tobytes(), float(d.Ps[isub]), mask[j].tobytes(),
                   float(tau_lin[j]), float(nu_fit_tau[j]))
            if key not in cache:
                ich = np.where(mask[j] > 0)[0]
                modelx = self._model_for(d.freqs[isub], nbin, d.Ps[isub], unscattered=fit_scat)[ich]
                if use_ird:
                    from .pptoaslib import instrumental_response_port_FT
                    irf = instrumental_response_port_FT(nbin, d.freqs[isub, ich], self.ird['DM'], d.Ps[isub],
                                                        self.ird['wids'], self.ird['irf_types'])
                    modelx = np.fft.irfft(irf * np.fft.rfft(modelx, axis=-1), axis=-1)
                mprof = modelx.mean(axis=0)
                if fit_scat:
                    k = np.arange(nbin // 2 + 1)
                    mprof = np.fft.irfft(np.fft.rfft(mprof) / (1.0 + 2.0j * np.pi * k * tau_lin[j]))
                cache[key] = mprof
            mprofs[j] = cache[key]
        return w, nu_mean, mprofs

    def _reference_phase_seeds(self, eng, port, freqs, P, w, nu_mean, mprofs, nu_fit_DM, DM_guess):
        """phi_guess of every subint exactly as the reference forms it (pptoas.py:421-457), in
        a pass of its own over the portraits: rot_prof = weighted mean over the good channels
        of the portrait dedispersed at DM_guess to their mean frequency (device: rotation +
        mean), fitted against the template's mean profile with fit_phase_shift(Ns=100) -- brute
        grid + SciPy's simplex finish retraced on the device --, then moved from nu_mean to
        nu_fit_DM.  (The route for batches without a single-pass path, Engine.fit_batch's
        ref_seed.)"""
        from .pplib import Dconst, phase_transform
        nok = len(P)
        # rotate_data(portx, 0.0, DM_guess, P, freqsx, nu_mean): the nu_mean term is the
        # same for every channel of a subint and rides on the phase argument; rotation,
        # weighted channel mean (channels of zero weight are not read) and the fit run
        # in one device call
        out = eng.reference_phase_seed(port, freqs, P, w, mprofs,
                                       phi=-Dconst * DM_guess / P * nu_mean ** -2.0,
                                       DM=np.full(nok, DM_guess), nu_DM=np.inf, Ns=100, finish='simplex')
        return np.array([phase_transform(out[j, 0], DM_guess, nu_mean[j], nu_fit_DM[j], P[j], mod=True)
                         for j in range(nok)])

    def get_TOAs(self, datafile=None, tscrunch=False, nu_refs=None, DM0=None,
                 bary=True, fit_DM=True, fit_GM=False, fit_scat=False,
                 log10_tau=True, scat_guess=None, fix_alpha=False,
                 print_phase=False, print_flux=False, print_parangle=False,
                 add_instrumental_response=False, addtnl_toa_flags={},
                 method='trust-ncg', bounds=None, nu_fits=None, show_plot=False,
                 quiet=None, seed='reference'):
        """Same arguments as the reference (pptoas.py:150-156), plus `seed`.  Not
        supported here: tscrunch and show_plot (they raise) -- they live in PSRCHIVE /
        the plotting code.

        seed='reference' (default -- a drop-in returns the reference's numbers): the
        reference's initial guesses, formed the way it forms them (pptoas.py:421-457:
        dedisperse to the mean frequency, weighted mean over the good channels,
        fit_phase_shift with SciPy's simplex finish retraced step for step, phase_transform
        to nu_fit) at the price of one more pass over the data; method='trust-ncg' then
        retraces SciPy's iteration from that very point, and get_TOAs returns the
        reference's own numbers, GM and scattering fits included.

        seed='device' (the fast path, ~2.3x the throughput): the phase seed is formed inside
        the fit (the exact maximum of the channel-summed cross-correlation on a pilot subset
        of the channels) and every `method` runs the Newton solver to the rounding of the
        objective -- the optimum itself, within ~1e-9 rot of wherever SciPy's iteration
        stops from the reference's own starting point."""
        if quiet is None:
            quiet = self.quiet
        if tscrunch or show_plot:
            raise NotImplementedError("tscrunch / plots are outside the accelerated path")
        if seed not in ('device', 'reference'):
            raise ValueError("seed must be 'device' or 'reference'")
        use_ird = bool(add_instrumental_response and
                       (self.ird['DM'] or len(self.ird['wids'])))
        if method not in ('trust-ncg', 'Newton-CG', 'TNC'):
            print("Method '%s' is not implemented." % method)
            sys.exit()
        self.nfit = 1 + int(fit_DM) + int(fit_GM) + 2 * int(fit_scat) - int(fix_alpha)
        self.fit_phi, self.fit_DM, self.fit_GM = True, fit_DM, fit_GM
        self.fit_tau = self.fit_alpha = fit_scat
        if fit_scat:
            self.fit_alpha = not fix_alpha
        self.fit_flags = [1, int(self.fit_DM), int(self.fit_GM), int(self.fit_tau),
                          int(self.fit_alpha)]
        self.log10_tau = log10_tau if fit_scat else False
        log10_tau = self.log10_tau
        if (self.fit_GM or fit_scat) and not quiet:
            print("You are using an experimental functionality of pptoas!")
        self.scat_guess = scat_guess
        self.DM0, self.bary = DM0, bary
        self.tscrunch = tscrunch
        self.add_instrumental_response = add_instrumental_response
        start = time.time()
        datafiles = self.datafiles if datafile is None else [datafile]
        eng = default_engine()
        last_fl = None          # (the reference's `fit_flags` variable lives across subints AND archives of one call)
        for iarch, datafile in enumerate(datafiles):
            try:
                data, fname = _load(datafile)
            except RuntimeError as err:
                if not quiet:
                    print("Cannot load_data(%s).  Skipping it." % datafile)
                    print(err)
                continue
            if not len(data.ok_isubs):
                if not quiet:
                    print("No subints to fit for %s.  Skipping it." % fname)
                continue
            self.ok_idatafiles.append(iarch)
            d = data
            nsub, nchan, nbin = d.nsub, d.nchan, d.nbin
            ok_isubs = np.asarray(d.ok_isubs, dtype=int)
            source = d.source if d.source is not None else "noname"
            obs = DataBunch(telescope=d.telescope, backend=d.backend, frontend=d.frontend)
            DM_stored = d.DM
            DM0_arch = DM_stored if self.DM0 is None else self.DM0
            MJDs = np.array([e.in_days() for e in d.epochs], dtype=np.double)

            # ---- batch marshalling: good subints, channel masks, templates ----
            nok = len(ok_isubs)
            mask = np.zeros((nok, nchan), dtype=np.uint8)
            nu_fit_arr = np.zeros((nok, 3))
            nu_ref_arr = np.full((nok, 3), np.nan)
            x0 = np.zeros((nok, 5))
            flags_per = []
            slots, slot_of = {}, np.zeros(nok, dtype=np.int32)
            errs = None if d.noise_stds is None else \
                np.ascontiguousarray(np.asarray(d.noise_stds)[ok_isubs, 0], dtype=np.float64)
            for j, isub in enumerate(ok_isubs):
                ich = np.asarray(d.ok_ichans[isub], dtype=int)
                mask[j, ich] = 1
                freqsx = d.freqs[isub, ich]
                key = d.freqs[isub].tobytes() + np.float64(d.Ps[isub]).tobytes() \
                    if (fit_scat or use_ird) else d.freqs[isub].tobytes()
                if use_ird:
                    key += ich.tobytes()    # the smearing width uses the good channels' spacing
                if key not in slots:
                    if len(slots) >= 64:
                        raise NotImplementedError("more than 64 distinct templates in one archive")
                    slots[key] = len(slots)
                    self._load_template(eng, slots[key], d.freqs[isub], nbin, d.Ps[isub],
                                        unscattered=fit_scat)
                    if use_ird:
                        # template x instrumental response of the good channels
                        # (pptoas.py:388-394), multiplied in the Fourier domain on the
                        # device: constant responses x per-channel dispersive smearing
                        from .pptoaslib import instrumental_response_device_args
                        rconst, smear = instrumental_response_device_args(
                            nbin, freqsx, self.ird['DM'], d.Ps[isub], self.ird['wids'],
                            self.ird['irf_types'], nchan=nchan, ichans=ich)
                        eng.apply_response(slots[key], rconst, smear)
                slot_of[j] = slots[key]
                if nu_fits is None:
                    nu_fit = guess_fit_freq(freqsx, d.SNRs[isub, 0, ich])
                    nu_fit_arr[j] = nu_fit
                else:
                    nu_fit_arr[j] = [nu_fits[0], nu_fits[0], nu_fits[-1]]
                if nu_refs is not None:
                    nu_ref_arr[j] = [nu_refs[0], nu_refs[0], nu_refs[-1]]
                    if bary and nu_refs[-1]:
                        nu_ref_arr[j, 2] = nu_refs[-1] / d.doppler_factors[isub]
                # initial guesses (pptoas.py:421-460); the phase comes from the
                # device seed
                tau_guess, alpha_guess = 0.0, 0.0
                if j == 0:
                    tau_lin = np.zeros(nok)      # tau_guess [rot] before any log10 (seed='reference')
                if fit_scat:
                    P = d.Ps[isub]
                    if self.scat_guess is not None:
                        tau_s, tau_ref, alpha_guess = self.scat_guess
                        tau_guess = (tau_s / P) * (nu_fit_arr[j, 2] / tau_ref) ** alpha_guess
                    else:
                        alpha_guess = self.alpha if hasattr(self, 'alpha') else scattering_alpha
                        if hasattr(self, 'gparams'):
                            tau_guess = (self.gparams[1] / P) * \
                                (nu_fit_arr[j, 2] / self.model_nu_ref) ** alpha_guess
                        else:
                            tau_guess = 0.0
                    tau_lin[j] = tau_guess
                    if log10_tau:
                        if tau_guess == 0.0:
                            tau_guess = nbin ** -1
                        tau_guess = np.log10(tau_guess)
                x0[j] = [0.0, DM_stored, 0.0, tau_guess, alpha_guess]
                if len(freqsx) == 1:
                    fl = [1, 0, 0, 0, 0]
                elif len(freqsx) == 2 and self.fit_DM and self.fit_GM:
                    # the reference writes `fit_flags[2] = 0` into the list LEFT OVER from the subint before
                    # (pptoas.py:479-481): right after a one-channel subint that list is [1,0,0,0,0] and the
                    # two-channel subint is fitted for phase only; after a normal one it becomes phase + DM.
                    # Reproduced -- but for the very first subint of a call, where the
                    # reference has no list yet and raises NameError: the flags asked for, GM dropped.
                    fl = list(last_fl) if last_fl is not None else list(self.fit_flags)
                    fl[2] = 0
                else:
                    fl = list(self.fit_flags)
                last_fl = fl
                flags_per.append(tuple(fl))
            port = _dededisperse(eng, _take_subints(d.subints, ok_isubs), d, ok_isubs)
            ref_in = None
            if seed == 'reference':
                # (batches without a single-pass path read the portraits twice -- seed, fit --:
                # hand them to the device once)
                port = _to_device_once(eng, port)
                ref_in = self._reference_seed_inputs(port, d, ok_isubs, mask, tau_lin, nu_fit_arr[:, 2],
                                                     fit_scat, use_ird)
            # ---- one device call per distinct flag set (normally one) ----
            res = None
            for fl in sorted(set(flags_per)):
                sel = np.array([k for k, f in enumerate(flags_per) if f == fl])
                # (all subints in one call is the normal case: no gather copy then)
                if len(sel) == nok:
                    psel = port
                elif hasattr(port, "is_cuda"):       # (device tensor: gather on the device)
                    import torch
                    psel = port[torch.as_tensor(sel, device=port.device)].contiguous()
                else:
                    psel = np.ascontiguousarray(port[sel])
                fkw = dict(errs=None if errs is None else errs[sel], nu_fits=nu_fit_arr[sel],
                           nu_outs=nu_ref_arr[sel], fit_flags=fl, log10_tau=log10_tau, option=0, is_toa=True,
                           model_slot=slot_of[sel], chan_mask=mask[sel], seed_ns=100 if seed == 'device' else 0,
                           method='newton' if seed == 'device' else method)
                fsel, Psel = d.freqs[ok_isubs][sel], np.asarray(d.Ps, dtype=np.float64)[ok_isubs][sel]
                r = None
                if ref_in is not None:
                    # the reference's own guess: formed inside the fit's single pass over the
                    # portraits where the library has that path, else in a pass of its own
                    w_, numean_, mprofs_ = (a[sel] for a in ref_in)
                    try:
                        r = eng.fit_batch(psel, fsel, Psel, x0[sel], ref_seed=dict(
                            weights=w_, model_profs=mprofs_, nu_mean=numean_, Ns=100, finish='simplex'), **fkw)
                    except EngineNotSupported:
                        x0[sel, 0] = self._reference_phase_seeds(eng, psel, fsel, Psel, w_, numean_, mprofs_,
                                                                 nu_fit_arr[sel, 0], DM_stored)
                if r is None:
                    r = eng.fit_batch(psel, fsel, Psel, x0[sel], **fkw)
                    if ref_in is not None:
                        # (the fallback route formed the guess in a pass of its own: report it like
                        # the single-pass route does, so every group carries the same keys)
                        r["seed_phase"] = x0[sel, 0].copy()
                if res is None:
                    res = {"duration": 0.0}
                for k, v in r.items():
                    if isinstance(v, np.ndarray):
                        # (groups may return different key sets: allocate on first sight)
                        if k not in res:
                            res[k] = np.zeros((nok,) + v.shape[1:], dtype=v.dtype)
                        res[k][sel] = v
                    elif k != "duration":
                        res.setdefault(k, v)
                res["duration"] += r["duration"]
            fit_duration = res["duration"]
            # template profile means per slot, for the flux estimate (the scattering
            # kernel leaves the mean of a profile unchanged: B_0 = 1)
            slot_means = {}
            if print_flux:
                for sl in set(slots.values()):
                    slot_means[sl] = eng.model_means(sl, nchan, nbin)

            # ---- TOA bookkeeping on the host (pptoas.py:528-721) ----
            phis = np.zeros(nsub); phi_errs = np.zeros(nsub)
            TOAs = np.zeros(nsub, dtype="object"); TOA_errs = np.zeros(nsub, dtype="object")
            DMs = np.zeros(nsub); DM_errs = np.zeros(nsub)
            GMs = np.zeros(nsub); GM_errs = np.zeros(nsub)
            taus = np.zeros(nsub); tau_errs = np.zeros(nsub)
            alphas = np.zeros(nsub); alpha_errs = np.zeros(nsub)
            scales = np.zeros([nsub, nchan]); scale_errs = np.zeros([nsub, nchan])
            snrs = np.zeros(nsub); channel_snrs = np.zeros([nsub, nchan])
            red_chi2s = np.zeros(nsub)
            profile_fluxes = np.zeros([nsub, nchan]); profile_flux_errs = np.zeros([nsub, nchan])
            fluxes = np.zeros(nsub); flux_errs = np.zeros(nsub); flux_freqs = np.zeros(nsub)
            covariances = np.zeros([nsub, self.nfit, self.nfit])
            nfevals = np.zeros(nsub, dtype="int"); rcs = np.zeros(nsub, dtype="int")
            nu_fits_out = list(np.zeros([nsub, 3])); nu_refs_out = list(np.zeros([nsub, 3]))
            for j, isub in enumerate(ok_isubs):
                fl = flags_per[j]
                P = d.Ps[isub]
                p, e = res["params"][j].copy(), res["param_errs"][j]
                ifit = np.where(fl)[0]
                cov = res["cov"][j][np.ix_(ifit, ifit)]
                TOA_MJD = d.epochs[isub] + MJD(0, (p[0] * P + d.backend_delay) / (3600 * 24.))
                TOA_err = e[0] * P * 1e6   # [us]
                df = d.doppler_factors[isub] if self.bary else 1.0
                DM_out, GM_out = p[1], p[2]
                if self.bary:
                    if fl[1]:
                        DM_out *= df
                    if fl[2]:
                        GM_out *= df ** 3
                nu_fits_out[isub] = list(nu_fit_arr[j])
                nu_refs_out[isub] = list(res["nu_refs"][j])
                phis[isub], phi_errs[isub] = p[0], e[0]
                TOAs[isub], TOA_errs[isub] = TOA_MJD, TOA_err
                DMs[isub], DM_errs[isub] = DM_out, e[1]
                GMs[isub], GM_errs[isub] = GM_out, e[2]
                taus[isub], tau_errs[isub] = p[3], e[3]
                alphas[isub], alpha_errs[isub] = p[4], e[4]
                nfevals[isub], rcs[isub] = res["nfeval"][j], res["return_code"][j]
                ich = np.asarray(d.ok_ichans[isub], dtype=int)
                scales[isub, ich] = res["scales"][j, ich]
                scale_errs[isub, ich] = res["scale_errs"][j, ich]
                snrs[isub] = res["snr"][j]
                channel_snrs[isub, ich] = res["channel_snrs"][j, ich]
                if cov.shape == covariances[isub].shape:
                    covariances[isub] = cov
                elif cov.shape == (1, 1):
                    # (`covariances[isub] = results.covariance_matrix`, pptoas.py:598: NumPy broadcasts a 1 x 1 matrix
                    # over the whole nfit x nfit slot instead of raising -- a phase-only fit fills every entry with
                    # var(phi); reproduced)
                    covariances[isub] = cov
                else:
                    for ii, a_ in enumerate(ifit):
                        for jj, b_ in enumerate(ifit):
                            if a_ < self.nfit and b_ < self.nfit:
                                covariances[isub][a_, b_] = cov[ii, jj]
                red_chi2s[isub] = res["red_chi2"][j]
                freqsx = d.freqs[isub, ich]
                if print_flux:       # pptoas.py:554-575
                    means = slot_means[slot_of[j]][ich]
                    profile_fluxes[isub, ich] = means * res["scales"][j, ich]
                    profile_flux_errs[isub, ich] = np.abs(means) * res["scale_errs"][j, ich]
                    fluxes[isub], flux_errs[isub] = weighted_mean(
                        profile_fluxes[isub, ich], profile_flux_errs[isub, ich])
                    flux_freqs[isub], _ = weighted_mean(freqsx, profile_flux_errs[isub, ich])
                toa_flags = {}
                DM_flag, DM_err_flag = (DM_out, e[1]) if fl[1] else (None, None)
                if fl[2]:
                    toa_flags['gm'] = GM_out
                    toa_flags['gm_err'] = e[2]
                if fl[3]:
                    if log10_tau:
                        toa_flags['scat_time'] = 10 ** p[3] * P / df * 1e6
                        toa_flags['log10_scat_time'] = p[3] + np.log10(P / df)
                        toa_flags['log10_scat_time_err'] = e[3]
                    else:
                        toa_flags['scat_time'] = p[3] * P / df * 1e6
                        toa_flags['scat_time_err'] = e[3] * P / df * 1e6
                    toa_flags['scat_ref_freq'] = res["nu_refs"][j, 2] * df
                    toa_flags['scat_ind'] = p[4]
                if fl[4]:
                    toa_flags['scat_ind_err'] = e[4]
                toa_flags['be'] = d.backend
                toa_flags['fe'] = d.frontend
                toa_flags['f'] = d.frontend + "_" + d.backend
                toa_flags['nbin'] = int(nbin)
                toa_flags['nch'] = int(nchan)
                toa_flags['nchx'] = int(len(freqsx))
                toa_flags['bw'] = freqsx.max() - freqsx.min()
                toa_flags['chbw'] = abs(d.bw) / nchan
                toa_flags['subint'] = int(isub)
                toa_flags['tobs'] = d.subtimes[isub]
                toa_flags['fratio'] = freqsx.max() / freqsx.min()
                toa_flags['tmplt'] = self.modelfile if isinstance(self.modelfile, str) \
                    else self.model_name
                toa_flags['snr'] = res["snr"][j]
                if nu_refs is not None and fl[0] and fl[1]:
                    toa_flags['phi_DM_cov'] = cov[0, 1]
                toa_flags['gof'] = res["red_chi2"][j]
                if print_phase:
                    toa_flags['phs'] = p[0]
                    toa_flags['phs_err'] = e[0]
                if print_flux:
                    toa_flags['flux'] = fluxes[isub]
                    toa_flags['flux_err'] = flux_errs[isub]
                    toa_flags['flux_ref_freq'] = flux_freqs[isub]
                if print_parangle:
                    toa_flags['par_angle'] = d.parallactic_angles[isub]
                for k, v in addtnl_toa_flags.items():
                    toa_flags[k] = v
                self.TOA_list.append(TOA(fname, res["nu_refs"][j, 0], TOA_MJD, TOA_err,
                                         d.telescope, d.telescope_code, DM_flag,
                                         DM_err_flag, toa_flags))
            # mean DM offset of the archive (pptoas.py:665-682)
            DeltaDMs = DMs - DM0_arch
            if np.all(DM_errs[ok_isubs]):
                DM_weights = DM_errs[ok_isubs] ** -2
            else:
                DM_weights = np.ones(len(ok_isubs))
            DeltaDM_mean, DeltaDM_var = np.average(DeltaDMs[ok_isubs], weights=DM_weights,
                                                   returned=True)
            DeltaDM_var = DeltaDM_var ** -1
            if len(ok_isubs) > 1:
                DeltaDM_var *= np.sum(((DeltaDMs[ok_isubs] - DeltaDM_mean) ** 2) *
                                      DM_weights) / (len(ok_isubs) - 1)
            self.order.append(fname)
            self.obs.append(obs)
            self.doppler_fs.append(d.doppler_factors)
            self.nu0s.append(d.nu0)
            self.nu_fits.append(nu_fits_out)
            self.nu_refs.append(nu_refs_out)
            self.ok_isubs.append(ok_isubs)
            self.epochs.append(d.epochs)
            self.MJDs.append(MJDs)
            self.Ps.append(d.Ps)
            self.phis.append(phis)
            self.phi_errs.append(phi_errs)
            self.TOAs.append(TOAs)
            self.TOA_errs.append(TOA_errs)
            self.DM0s.append(DM0_arch)
            self.DMs.append(DMs)
            self.DM_errs.append(DM_errs)
            self.DeltaDM_means.append(DeltaDM_mean)
            self.DeltaDM_errs.append(DeltaDM_var ** 0.5)
            self.GMs.append(GMs)
            self.GM_errs.append(GM_errs)
            self.taus.append(taus)
            self.tau_errs.append(tau_errs)
            self.alphas.append(alphas)
            self.alpha_errs.append(alpha_errs)
            self.scales.append(scales)
            self.scale_errs.append(scale_errs)
            self.snrs.append(snrs)
            self.channel_snrs.append(channel_snrs)
            self.profile_fluxes.append(profile_fluxes)
            self.profile_flux_errs.append(profile_flux_errs)
            self.fluxes.append(fluxes)
            self.flux_errs.append(flux_errs)
            self.flux_freqs.append(flux_freqs)
            self.covariances.append(covariances)
            self.red_chi2s.append(red_chi2s)
            self.nfevals.append(nfevals)
            self.rcs.append(rcs)
            self.fit_durations.append(fit_duration)
            if not quiet:
                print("--------------------------")
                print(fname)
                print("~%.6f sec/TOA" % (fit_duration / len(ok_isubs)))
                print("Med. TOA error is %.3f us" % (np.median(phi_errs[ok_isubs]) *
                                                     d.Ps.mean() * 1e6))
        tot_duration = time.time() - start
        if not quiet and len(self.ok_isubs):
            print("--------------------------")
            print("Total time: %.2f sec, ~%.4f sec/TOA" %
                  (tot_duration, tot_duration / sum(len(o) for o in self.ok_isubs)))

    def get_narrowband_TOAs(self, datafile=None, tscrunch=False, fit_scat=False,
                            log10_tau=True, scat_guess=None, print_phase=False,
                            print_flux=False, print_parangle=False,
                            add_instrumental_response=False, addtnl_toa_flags={},
                            method='trust-ncg', bounds=None, show_plot=False, quiet=None):
        """One TOA per channel (pptoas.py:744-1120): every good channel of every
        good subint is fitted for a phase shift and an amplitude against its
        template profile -- fit_phase_shift, bounds [-0.5, 0.5], Ns = 100 -- in one
        device batch per archive.  As in the reference, scattering fits are not
        implemented here; print_phase / print_flux cannot work in the reference's
        narrowband path (it reads fields fit_phase_shift does not return) and raise."""
        if quiet is None:
            quiet = self.quiet
        if fit_scat or tscrunch or show_plot or print_phase or print_flux:
            raise NotImplementedError("fit_scat / tscrunch / show_plot / print_phase / print_flux "
                                      "are not available for narrowband TOAs")
        if not quiet:
            print("You are using an experimental functionality of pptoas!")
        self.nfit = 1
        self.fit_phi, self.fit_tau = True, False
        self.fit_flags = [1, 0]
        self.log10_tau = False
        self.scat_guess = scat_guess
        self.tscrunch = tscrunch
        self.add_instrumental_response = add_instrumental_response
        use_ird = bool(add_instrumental_response and
                       (self.ird['DM'] or len(self.ird['wids'])))
        start = time.time()
        datafiles = self.datafiles if datafile is None else [datafile]
        eng = default_engine()
        for iarch, datafile in enumerate(datafiles):
            try:
                d, fname = _load(datafile)
            except RuntimeError:
                if not quiet:
                    print("Cannot load_data(%s).  Skipping it." % datafile)
                continue
            if not len(d.ok_isubs):
                if not quiet:
                    print("No subints to fit for %s.  Skipping it." % fname)
                continue
            self.ok_idatafiles.append(iarch)
            nsub, nchan, nbin = d.nsub, d.nchan, d.nbin
            ok_isubs = np.asarray(d.ok_isubs, dtype=int)
            obs = DataBunch(telescope=d.telescope, backend=d.backend, frontend=d.frontend)
            z2 = lambda dt=np.float64: np.zeros([nsub, nchan], dtype=dt)  # noqa: E731
            phis, phi_errs, taus, tau_errs = z2(), z2(), z2(), z2()
            TOAs, TOA_errs = z2("object"), z2("object")
            scales, scale_errs, channel_snrs = z2(), z2(), z2()
            profile_fluxes, profile_flux_errs, channel_red_chi2s = z2(), z2(), z2()
            covariances = np.zeros([nsub, nchan, self.nfit, self.nfit])
            nfevals, rcs = z2("int"), z2("int")
            MJDs = np.array([e.in_days() for e in d.epochs], dtype=np.double)
            # ---- gather every (subint, good channel) profile pair ----
            profs, mprofs, noises, where = [], [], [], []
            sub_all = np.asarray(d.subints)
            if d.dmc:
                sub_all = sub_all.copy()
                sub_all[ok_isubs, 0] = _dededisperse(eng, _take_subints(sub_all, ok_isubs), d, ok_isubs)
            for isub in ok_isubs:
                ich = np.asarray(d.ok_ichans[isub], dtype=int)
                model = np.asarray(self._model_for(d.freqs[isub], nbin, d.Ps[isub]))
                modelx = model[ich]
                if use_ird:
                    from .pptoaslib import instrumental_response_port_FT
                    resp = instrumental_response_port_FT(nbin, d.freqs[isub, ich], self.ird['DM'],
                                                         d.Ps[isub], self.ird['wids'],
                                                         self.ird['irf_types'])
                    modelx = np.fft.irfft(resp * np.fft.rfft(modelx, axis=-1), axis=-1)
                profs.append(sub_all[isub, 0, ich])
                mprofs.append(modelx)
                # (NaN: the device measures the noise from the power spectrum)
                noises.append(np.full(len(ich), np.nan) if d.noise_stds is None
                              else np.asarray(d.noise_stds)[isub, 0, ich])
                where += [(isub, ichan) for ichan in ich]
            t0 = time.time()
            out = eng.fit_phase_shift_batch(np.concatenate(profs), np.concatenate(mprofs),
                                            np.concatenate(noises).astype(np.float64),
                                            bounds=(-0.5, 0.5), Ns=100, finish='simplex')
            fit_duration = time.time() - t0
            # ---- TOA bookkeeping (pptoas.py:994-1088) ----
            for (isub, ichan), r in zip(where, out):
                phase, phase_err, scale, scale_err, snr, red_chi2 = r[:6]
                P = d.Ps[isub]
                TOA_MJD = d.epochs[isub] + MJD(0, (phase * P + d.backend_delay) / (3600 * 24.))
                TOA_err = phase_err * P * 1e6
                phis[isub, ichan], phi_errs[isub, ichan] = phase, phase_err
                TOAs[isub, ichan], TOA_errs[isub, ichan] = TOA_MJD, TOA_err
                scales[isub, ichan], scale_errs[isub, ichan] = scale, scale_err
                channel_snrs[isub, ichan] = snr
                channel_red_chi2s[isub] = red_chi2     # (the reference assigns the whole row)
                toa_flags = {'be': d.backend, 'fe': d.frontend, 'f': d.frontend + "_" + d.backend,
                             'nbin': int(nbin), 'bw': abs(d.bw) / nchan, 'subint': int(isub),
                             'chan': int(ichan), 'tobs': d.subtimes[isub],
                             'tmplt': self.modelfile if isinstance(self.modelfile, str)
                             else self.model_name, 'snr': snr, 'gof': red_chi2}
                if print_parangle:
                    toa_flags['par_angle'] = d.parallactic_angles[isub]
                for k, v in addtnl_toa_flags.items():
                    toa_flags[k] = v
                self.TOA_list.append(TOA(fname, d.freqs[isub, ichan], TOA_MJD, TOA_err,
                                         d.telescope, d.telescope_code, None, None, toa_flags))
            self.order.append(fname)
            self.obs.append(obs)
            self.doppler_fs.append(d.doppler_factors)
            self.ok_isubs.append(ok_isubs)
            self.epochs.append(d.epochs)
            self.MJDs.append(MJDs)
            self.Ps.append(d.Ps)
            self.phis.append(phis)
            self.phi_errs.append(phi_errs)
            self.TOAs.append(TOAs)
            self.TOA_errs.append(TOA_errs)
            self.taus.append(taus)
            self.tau_errs.append(tau_errs)
            self.scales.append(scales)
            self.scale_errs.append(scale_errs)
            self.channel_snrs.append(channel_snrs)
            self.profile_fluxes.append(profile_fluxes)
            self.profile_flux_errs.append(profile_flux_errs)
            self.covariances.append(covariances)
            self.channel_red_chi2s.append(channel_red_chi2s)
            self.nfevals.append(nfevals)
            self.rcs.append(rcs)
            self.fit_durations.append(fit_duration)
            if not quiet:
                print("--------------------------")
                print(fname)
                print("~%.4f sec/TOA" % (fit_duration / max(1, len(self.TOA_list))))
                print("Med. TOA error is %.3f us" % (np.median(phi_errs[ok_isubs]) *
                                                     d.Ps.mean() * 1e6))
        tot_duration = time.time() - start
        if not quiet and len(self.ok_isubs):
            print("--------------------------")
            print("Total time: %.2f sec, ~%.4f sec/TOA" %
                  (tot_duration, tot_duration / max(1, len(self.TOA_list))))

    def get_channels_to_zap(self, SNR_threshold=8.0, rchi2_threshold=1.3, iterate=True,
                            show=False):
        """Flag channels by per-channel reduced chi^2 and S/N (pptoas.py:1208-1285;
        get_TOAs must have been called first).  The reduced chi^2 of every channel of
        every fitted subint -- rotated data minus scaled template in the time domain,
        dof = nbin - 2, as show_fit / get_red_chi2 form it -- comes from one device
        pass per archive; the threshold logic is the reference's.  Appends to
        self.channel_red_chi2s and self.zap_channels (one entry per archive, each a
        list over its fitted subints)."""
        if show:
            raise NotImplementedError("plots are outside the accelerated path")
        eng = default_engine()
        for iarch, ok_idatafile in enumerate(self.ok_idatafiles):
            data, fname = _load(self.datafiles[ok_idatafile])
            d = data
            nbin = d.nbin
            ok_isubs = np.asarray(self.ok_isubs[iarch], dtype=int)
            nok = len(ok_isubs)
            params = np.zeros((nok, 5))
            slots, slot_of = {}, np.zeros(nok, dtype=np.int32)
            scales = np.zeros((nok, d.nchan))
            for j, isub in enumerate(ok_isubs):
                df = self.doppler_fs[iarch][isub] if self.bary else 1.0
                tau = self.taus[iarch][isub]
                if tau != 0.0 and self.log10_tau:
                    tau = 10.0 ** tau
                params[j] = [self.phis[iarch][isub], self.DMs[iarch][isub] / df,
                             self.GMs[iarch][isub] / df ** 3, tau, self.alphas[iarch][isub]]
                scat = bool(tau != 0.0)
                use_ird = bool(getattr(self, "add_instrumental_response", False) and
                               (self.ird['DM'] or len(self.ird['wids'])))
                key = (d.freqs[isub].tobytes(), scat,
                       np.float64(d.Ps[isub]).tobytes() if use_ird else b"")
                if key not in slots:
                    if len(slots) >= 64:
                        raise NotImplementedError("more than 64 distinct templates in one archive")
                    slots[key] = len(slots)
                    self._load_template(eng, slots[key], d.freqs[isub], nbin, d.Ps.mean(),
                                        unscattered=scat)
                    if use_ird:     # show_fit applies it over all channels (pptoas.py:1389-1395)
                        from .pptoaslib import instrumental_response_device_args
                        rconst, smear = instrumental_response_device_args(
                            nbin, d.freqs[isub], self.ird['DM'], d.Ps[isub], self.ird['wids'],
                            self.ird['irf_types'])
                        eng.apply_response(slots[key], rconst, smear)
                slot_of[j] = slots[key]
                scales[j] = self.scales[iarch][isub]
            port = _dededisperse(eng, _take_subints(d.subints, ok_isubs), d, ok_isubs)
            noise = _noise_rows(d, ok_isubs)
            nu_refs = np.array([self.nu_refs[iarch][isub] for isub in ok_isubs], dtype=np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                rchi2 = eng.channel_red_chi2(port, d.freqs[ok_isubs], d.Ps[ok_isubs], params,
                                             nu_refs, scales, noise, slots=slot_of)
            channel_red_chi2s, zap_channels = [], []
            for j, isub in enumerate(ok_isubs):
                ok_ichans = [int(v) for v in d.ok_ichans[isub]]
                channel_snrs = self.channel_snrs[iarch][isub]
                thresh = (SNR_threshold ** 2.0 / len(ok_ichans)) ** 0.5
                red_chi2s, bad = [], []
                for ichan in ok_ichans:
                    r = rchi2[j, ichan]
                    red_chi2s.append(r)
                    if r > rchi2_threshold or np.isnan(r):
                        bad.append(ichan)
                    elif SNR_threshold and channel_snrs[ichan] < thresh:
                        bad.append(ichan)
                if iterate and SNR_threshold and len(bad):
                    old_len, added_new = len(bad), True
                    while added_new and (len(ok_ichans) - len(bad)):
                        thresh = (SNR_threshold ** 2.0 / (len(ok_ichans) - len(bad))) ** 0.5
                        for ichan in ok_ichans:
                            if ichan not in bad and channel_snrs[ichan] < thresh:
                                bad.append(ichan)
                        added_new = bool(len(bad) - old_len)
                        old_len = len(bad)
                channel_red_chi2s.append(red_chi2s)
                zap_channels.append(bad)
            self.channel_red_chi2s.append(channel_red_chi2s)
            self.zap_channels.append(zap_channels)
