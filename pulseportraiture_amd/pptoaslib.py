"""Reference-shaped wideband fit: `fit_portrait_full` (pptoaslib.py:928-1096)
and its batched form.  The arithmetic (rFFT, cross-spectrum, chi^2 surface,
trust-region solve, zero-covariance frequencies, covariance with amplitudes)
runs in the HIP kernels of csrc/ via the C ABI."""
import sys

import numpy as np

from .engine import default_engine
from .pplib import DataBunch, Dconst, RCSTRINGS  # noqa: F401

_METHODS = ('trust-ncg', 'Newton-CG', 'TNC')


def _bunch(res, i, fit_flags):
    ifit = np.where(fit_flags)[0]
    p, e = res["params"][i], res["param_errs"][i]
    cov = res["cov"][i][np.ix_(ifit, ifit)]
    return DataBunch(
        params=list(p), param_errs=e.copy(), phi=p[0], phi_err=e[0], DM=p[1],
        DM_err=e[1], GM=p[2], GM_err=e[2], tau=p[3], tau_err=e[3], alpha=p[4],
        alpha_err=e[4], scales=res["scales"][i].copy(),
        scale_errs=res["scale_errs"][i].copy(), nu_DM=res["nu_refs"][i, 0],
        nu_GM=res["nu_refs"][i, 1], nu_tau=res["nu_refs"][i, 2],
        covariance_matrix=cov, chi2=res["chi2"][i], red_chi2=res["red_chi2"][i],
        snr=res["snr"][i], channel_snrs=res["channel_snrs"][i].copy(),
        duration=res["duration"], nfeval=int(res["nfeval"][i]),
        return_code=int(res["return_code"][i]),
        # (not a field of the reference's result: how many of the nfeval evaluations were
        # passes over the data)
        npass=int(res["npass"][i]))


def fit_portrait_full(data_port, model_port, init_params, P, freqs,
                      nu_fits=[None, None, None], nu_outs=[None, None, None],
                      errs=None, fit_flags=[1, 1, 1, 1, 1],
                      bounds=[(None, None), (None, None), (None, None),
                              (None, None), (None, None)], log10_tau=True,
                      option=0, sub_id=None, method='trust-ncg', is_toa=True,
                      quiet=True):
    """Fit phase, DM, GM, tau and alpha between a data and a model portrait.

    Same arguments and result fields as the reference.  method='trust-ncg' (the
    reference's default) walks SciPy's trust-ncg iteration on the device and stops
    where the reference stops; 'Newton-CG' and 'TNC' both run the device's Newton
    solver to the rounding of the objective.  `bounds` are honoured for 'TNC' only,
    like the reference (pptoaslib.py:995-997 drops them for the other methods): the
    box-constrained optimum is found by an active-set iteration over device fits
    (_fit_with_bounds); an unknown method exits like the reference does
    (pptoaslib.py:1008-1010)."""
    if method not in _METHODS:
        print("Method '%s' is not implemented." % method)
        sys.exit()
    eng = default_engine()
    flags = [1 if f else 0 for f in fit_flags]
    eng.set_model_cached(model_port, slot=0)
    data = np.asarray(data_port)
    if method == 'TNC' and bounds is not None and any(
            b is not None and (b[0] is not None or b[1] is not None) for b in bounds):
        res = _fit_with_bounds(eng, data[None] if data.ndim == 2 else data, freqs, P, init_params,
                               errs, list(nu_fits), list(nu_outs), flags, bounds, log10_tau, option,
                               is_toa)
        return _bunch(res, 0, flags)
    res = eng.fit_batch(data[None] if data.ndim == 2 else data, freqs, P,
                        init_params, errs=errs, nu_fits=[list(nu_fits)],
                        nu_outs=[list(nu_outs)], fit_flags=flags,
                        log10_tau=log10_tau, option=option, is_toa=is_toa,
                        method=method)
    r = _bunch(res, 0, flags)
    if r.return_code not in (0, 1, 2, 4):
        rcs = "NaN or singular objective"
        if sub_id is not None:
            ii = sub_id[::-1].index("_")
            sys.stderr.write("Fit 'failed' with return code %d: %s -- %s subint %s\n"
                             % (r.return_code, rcs, sub_id[:-ii - 1], sub_id[-ii:]))
        else:
            sys.stderr.write("Fit 'failed' with return code %d -- %s" %
                             (r.return_code, rcs))
    return r


def _fit_with_bounds(eng, data, freqs, P, init_params, errs, nu_fits, nu_outs, flags, bounds,
                     log10_tau, option, is_toa):
    """method='TNC' with finite bounds (pptoaslib.py:995-1007: the only method the
    reference applies them for).  The box applies to the parameters AT the fit's
    reference frequencies, as in the reference.  Active-set iteration, every step one
    device fit (Newton solver): parameters found outside their bounds are fixed at the
    bound and the rest refitted; a fixed parameter whose gradient points back into the
    box is released.  The post-fit quantities (zero-covariance frequencies, errors,
    covariance, scales) are then taken at that point with the caller's fit_flags, as
    the reference takes them at whatever its minimiser returns."""
    lo = np.array([-np.inf if (b is None or b[0] is None) else float(b[0]) for b in bounds])
    hi = np.array([np.inf if (b is None or b[1] is None) else float(b[1]) for b in bounds])
    if np.any(lo > hi):
        raise ValueError("bounds: lower > upper")
    x = np.clip(np.asarray(init_params, dtype=np.float64).copy(), lo, hi)   # (TNC starts inside the box)
    # The iteration works with RAW parameters -- phi, tau at the fit's own reference frequencies --
    # so those must be concrete before the loop: a None entry means "the mean channel frequency"
    # for the fit (pptoaslib.py:986-989) but "the zero-covariance frequency" as an OUTPUT frequency,
    # and a phase returned there would be fed back as a guess at nu_fit.
    fmean = float(np.mean(np.asarray(freqs, dtype=np.float64)))
    nu_fits = [fmean if (v is None or (isinstance(v, float) and np.isnan(v))) else float(v) for v in nu_fits]
    common = dict(errs=errs, nu_fits=[nu_fits], log10_tau=log10_tau, option=option, is_toa=is_toa)
    active = {}
    max_iter = eng.get_option("max_iter")       # (the caller's setting: restored after the evaluate-only calls)
    nfeval, duration = 0, 0.0
    for _ in range(12):
        fl = [1 if (f and j not in active) else 0 for j, f in enumerate(flags)]
        for j, v in active.items():
            x[j] = v
        if any(fl):
            # raw parameters: output frequencies = the fit's own
            r = eng.fit_batch(data, freqs, P, x, nu_outs=[nu_fits], fit_flags=fl, method='newton', **common)
            nfeval += int(r["nfeval"][0])
            duration += r["duration"]
            xs = r["params"][0].copy()
            xs[0] = x[0] + ((xs[0] - x[0] + 0.5) % 1.0 - 0.5)     # (the phase comes back wrapped)
        else:
            xs = x.copy()
        out = [j for j in range(5) if fl[j] and not (lo[j] <= xs[j] <= hi[j])]
        if out:
            j = max(out, key=lambda q: max(lo[q] - xs[q], xs[q] - hi[q]) / (abs(xs[q]) + 1e-300))
            active[j] = lo[j] if xs[j] < lo[j] else hi[j]
            x = np.clip(xs, lo, hi)
            continue
        x = xs
        # gradient of the full problem at x: may a fixed parameter move back inside?
        eng.set_option("max_iter", 0)
        try:
            r = eng.fit_batch(data, freqs, P, x, nu_outs=[nu_fits], fit_flags=flags, objective=True,
                              method='newton', **common)
        finally:
            eng.set_option("max_iter", max_iter)
        g = r["obj_grad"][0]
        nfeval += int(r["nfeval"][0])
        duration += r["duration"]
        free = [j for j, v in active.items() if (v == lo[j] and g[j] < 0.0) or (v == hi[j] and g[j] > 0.0)]
        if not free:
            break
        for j in free:
            del active[j]
    # post-fit stage at x with the caller's flags and output frequencies
    eng.set_option("max_iter", 0)
    try:
        res = eng.fit_batch(data, freqs, P, x, nu_outs=[nu_outs], fit_flags=flags, method='newton', **common)
    finally:
        eng.set_option("max_iter", max_iter)
    res["return_code"][:] = 2 if not active else 0     # (TNC's table: XCONVERGED / LOCALMINIMUM at a bound)
    # what the whole bounded fit cost, not its closing evaluate-only call
    res["nfeval"][:] = nfeval + int(res["nfeval"][0])
    res["duration"] = duration + res["duration"]
    return res


def rotate_portrait_full(port, phi, DM, GM, freqs, nu_DM=np.inf, nu_GM=np.inf, P=None):
    """Rotate / dedisperse a portrait by (phi, DM, GM) on the GPU
    (pptoaslib.py:52-81)."""
    if P is None:
        P = 1.0
    port = np.asarray(port, dtype=np.float64)
    return default_engine().rotate_portraits(port[None], freqs, P, phi=phi, DM=DM, GM=GM,
                                             nu_DM=nu_DM, nu_GM=nu_GM)[0]


def fit_portrait_full_batch(data_ports, model_port, init_params, Ps, freqs,
                            nu_fits=None, nu_outs=None, errs=None,
                            fit_flags=[1, 1, 0, 0, 0], log10_tau=False, option=0,
                            is_toa=True, chan_mask=None, model_slot=None,
                            engine=None, method='trust-ncg'):
    """Batched form: data_ports[nsub,nchan,nbin] against one model (or several
    pre-loaded slots); returns the dict of result arrays of Engine.fit_batch."""
    eng = engine or default_engine()
    if model_port is not None:
        eng.set_model_cached(model_port, slot=0)
    return eng.fit_batch(data_ports, freqs, Ps, init_params, errs=errs,
                         nu_fits=nu_fits, nu_outs=nu_outs, fit_flags=fit_flags,
                         log10_tau=log10_tau, option=option, is_toa=is_toa,
                         chan_mask=chan_mask, model_slot=model_slot, method=method)


# --------------------------------------------------------------------------
# instrumental response applied to the template (host-side input preparation of
# get_TOAs(add_instrumental_response=True); pptoaslib.py:14-50, 112-179)
# --------------------------------------------------------------------------
def gaussian_profile_FT(nbin, loc, wid, amp):
    """Analytic Fourier transform of a Gaussian profile of FWHM wid [rot] at
    phase loc, sampled at nbin/2 + 1 harmonics, windowing included through the
    sinc-Gauss convolution formula (pptoaslib.py:14-50)."""
    from scipy.special import erf
    nharm = nbin // 2 + 1
    if wid <= 0.0:
        return np.zeros(nharm, 'd')
    sigma = wid / (2 * np.sqrt(2 * np.log(2)))
    amp = amp * (2 * np.pi * sigma ** 2) ** 0.5
    sigma = 1.0 / (sigma * 2 * np.pi)
    k = np.arange(nharm)
    a = sigma / ((1.0 / np.pi) * 2 ** 0.5)
    b = k / (sigma * 2 ** 0.5)
    vals = np.exp(-b ** 2) * (erf(a - b * 1j) + erf(a + b * 1j)) / 2
    vals = vals * (amp * nbin)
    if loc != 0.0:
        vals = vals * np.exp(-k * 2.0j * np.pi * loc)
    return np.nan_to_num(vals)


def instrumental_response_FT(nbin, wid=0.0, irf_type='rect'):
    """Fourier transform of one instrumental response of width wid [rot]: a
    rectangle ('rect', a sinc) or a Gaussian of that FWHM ('gauss')
    (pptoaslib.py:112-143)."""
    nharm = nbin // 2 + 1
    if wid == 0.0:
        return np.ones(nharm)
    if irf_type == 'rect':
        return np.sinc(np.arange(nharm) * wid)
    if irf_type == 'gauss':
        g = gaussian_profile_FT(nbin, 0.0, wid, 1.0)
        return g / g[0]
    print("Unrecognized instrumental response function type '%s'." % irf_type)
    return 0


def instrumental_response_port_FT(nbin, freqs, DM=0.0, P=1.0, wids=[], irf_types=[]):
    """Combined response per channel: the constant responses times the
    dispersive smearing of DM across a channel, 8.3e-6 chan_bw / nu_GHz^3 / P
    rotations wide (pptoaslib.py:145-179)."""
    nharm = nbin // 2 + 1
    nchan = len(freqs)
    if DM == len(wids) == 0.0:
        return np.ones([nchan, nharm])
    resp = np.ones([nchan, nharm], dtype=np.complex128)
    for wid, irf_type in zip(wids, irf_types):
        resp *= np.asarray(instrumental_response_FT(nbin, wid, irf_type))[None, :]
    if DM:
        chan_bw = abs(freqs[1] - freqs[0])
        for ichan, freq in enumerate(freqs):
            wid = 8.3e-6 * chan_bw / (freq / 1e3) ** 3 / P
            resp[ichan] *= instrumental_response_FT(nbin, wid, 'rect')
    return resp


def instrumental_response_device_args(nbin, freqs, DM=0.0, P=1.0, wids=[], irf_types=[], nchan=None,
                                      ichans=None):
    """The two factors of instrumental_response_port_FT as Engine.apply_response takes
    them: the product of the constant responses (nbin/2 + 1 complex values, or None)
    and the dispersive smearing width of every channel in rotations (or None).
    `freqs` are the channels the response applies to (pptoas.py:388-394 passes the good
    channels: the smearing uses THEIR spacing); with ichans / nchan the widths are
    scattered into a full-length array, zero (no response) elsewhere."""
    nharm = nbin // 2 + 1
    rconst = None
    if len(wids):
        rconst = np.ones(nharm, dtype=np.complex128)
        for wid, irf_type in zip(wids, irf_types):
            rconst = rconst * np.asarray(instrumental_response_FT(nbin, wid, irf_type))
    smear = None
    if DM:
        freqs = np.asarray(freqs, dtype=np.float64)
        chan_bw = abs(freqs[1] - freqs[0])
        w = 8.3e-6 * chan_bw / (freqs / 1e3) ** 3 / P
        if ichans is not None:
            smear = np.zeros(int(nchan))
            smear[np.asarray(ichans, dtype=int)] = w
        else:
            smear = w
    return rconst, smear
