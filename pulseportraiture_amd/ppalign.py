"""Iterative alignment and averaging of subintegrations: the array form of
ppalign.align_archives' loop (reference ppalign.py:110-214).

Each subint is fitted for a phase, a DM and channel amplitudes against the
current template -- the reference's own iteration (ppalign.py:180-195): its phase guess
(fit_phase_shift of the dedispersed channel mean, Ns = nbin, SciPy's simplex finish retraced
on the device) and SciPy's trust-ncg retraced from that point, one batched call each --, rotated by the fit and
added to the average with weights scales / errs**2 (one device pass,
Engine.align_accumulate); the average becomes the template of the next iteration.
PSRFITS I/O (load_data / unload_new_archive) stays outside: portraits come in as
arrays.
"""
import numpy as np

from .engine import default_engine
from .pplib import guess_fit_freq, fit_phase_shift, get_noise


def normalize_portrait(port, method="rms", weights=None, return_norms=False):
    """Normalise every profile of a portrait (pplib.py:2462-2507)."""
    if method not in ("mean", "max", "prof", "rms", "abs"):
        print("Unknown method for normalize_portrait(...), '%s'." % method)
        return None
    port = np.asarray(port, dtype=np.float64)
    norm_port = np.zeros(port.shape)
    norm_vals = np.ones(len(port))
    if method == "prof":
        good = np.where(port.sum(axis=1) != 0.0)[0]
        w = np.ones(len(good)) if weights is None else np.asarray(weights)[good]
        mean_prof = np.average(port[good], axis=0, weights=w)
    for ichan in range(len(port)):
        if port[ichan].any():
            if method == "mean":
                norm = port[ichan].mean()
            elif method == "max":
                norm = port[ichan].max()
            elif method == "prof":
                norm = fit_phase_shift(port[ichan], mean_prof).scale
            elif method == "rms":
                norm = get_noise(port[ichan])
            else:
                norm = (port[ichan] ** 2.0).sum() ** 0.5
            norm_port[ichan] = port[ichan] / norm
            norm_vals[ichan] = norm
    return (norm_port, norm_vals) if return_norms else norm_port


def align_subints(ports, freqs, Ps, noise_stds, model_port, weights=None, SNRs=None,
                  DM_guess=0.0, fit_dm=True, niter=1, norm=None, engine=None,
                  return_fits=False, quiet=True):
    """Align and average subints against an initial template.

    ports[nsub,nchan,nbin], freqs[nchan] or [nsub,nchan], Ps[nsub],
    noise_stds[nsub,nchan], model_port[nchan,nbin] (the initial guess, same
    channels: the reference's same_freqs branch), weights[nsub,nchan] (0 = channel
    not usable in that subint), SNRs[nsub,nchan] for guess_fit_freq, DM_guess the
    header DM of non-dedispersed data (0.0 if dedispersed).

    Returns the aligned [nchan,nbin] average (channels never hit stay zero), and
    with return_fits=True also the last iteration's fit results.
    """
    eng = engine or default_engine()
    ports = np.asarray(ports)
    nsub, nchan, nbin = ports.shape
    freqs = np.asarray(freqs, dtype=np.float64)
    f2 = freqs if freqs.ndim == 2 else np.broadcast_to(freqs, (nsub, nchan))
    Ps = np.broadcast_to(np.asarray(Ps, dtype=np.float64), (nsub,)).copy()
    errs = np.asarray(noise_stds, dtype=np.float64)
    wts = np.ones((nsub, nchan)) if weights is None else np.asarray(weights, dtype=np.float64)
    snrs = np.ones((nsub, nchan)) if SNRs is None else np.asarray(SNRs, dtype=np.float64)
    mask = (wts > 0.0).astype(np.uint8)
    model_port = np.asarray(model_port, dtype=np.float64)
    from .pplib import Dconst
    res = None
    multi = mask.sum(axis=1) > 1          # (subints with one usable channel: the reference's 1-channel hack)
    single = mask.sum(axis=1) == 1
    for it in range(int(niter)):
        if not quiet:
            print("Doing iteration %d..." % (it + 1))
        eng.set_model(model_port)
        nu_fit = np.array([guess_fit_freq(f2[i][mask[i] > 0], snrs[i][mask[i] > 0])
                           if mask[i].any() else f2[i].mean() for i in range(nsub)])
        # ---- the reference's own iteration (ppalign.py:180-195) ----
        # phase_guess = fit_phase_shift(average(rotate_data(port, 0, DM_guess, P, freqs, nu_fit), axis=0,
        #                                       weights=weights[ichans]), model[ichans].mean(axis=0), Ns=nbin).phase:
        # rotation to nu_fit, weighted channel mean and the fit (brute grid of nbin points + SciPy's simplex
        # finish retraced) in one device call; the template's mean profile is taken over the channels the
        # subint uses.  Neither wrapped nor moved to another frequency (the rotation is about nu_fit already).
        mprofs = np.empty((nsub, nbin))
        cache = {}
        for i in range(nsub):
            key = mask[i].tobytes()
            if key not in cache:
                ich = np.where(mask[i] > 0)[0]
                cache[key] = model_port[ich].mean(axis=0) if len(ich) else np.zeros(nbin)
            mprofs[i] = cache[key]
        x0 = np.zeros((nsub, 5))
        x0[:, 1] = DM_guess
        phase, DM, nu_ref = np.zeros(nsub), np.full(nsub, float(DM_guess)), nu_fit.copy()
        scales = np.zeros((nsub, nchan))
        isel = np.where(multi)[0]
        if len(isel):
            take = (lambda a: a) if len(isel) == nsub else (lambda a: np.ascontiguousarray(a[isel]))
            seed = eng.reference_phase_seed(take(ports), take(f2), take(Ps), take(np.where(mask > 0, wts, 0.0)),
                                            take(mprofs), phi=take(-Dconst * DM_guess / Ps * nu_fit ** -2.0),
                                            DM=np.full(len(isel), float(DM_guess)), nu_DM=np.inf, Ns=nbin,
                                            finish='simplex')
            x0[isel, 0] = seed[:, 0]
            flags = [1, int(bool(fit_dm)), 0, 0, 0]
            # fit_portrait_full(port, model, [phase_guess, DM_guess, 0, 0, 0], P, freqs, [nu_fit] * 3, [None] * 3,
            #                   errs, fit_flags, log10_tau=False): SciPy's trust-ncg retraced from that very point
            res = eng.fit_batch(take(ports), take(f2), take(Ps), take(x0), errs=take(errs), chan_mask=take(mask),
                                nu_fits=np.repeat(take(nu_fit)[:, None], 3, axis=1), fit_flags=flags,
                                log10_tau=False, method='trust-ncg')
            phase[isel], DM[isel], nu_ref[isel] = res["params"][:, 0], res["params"][:, 1], res["nu_refs"][:, 0]
            scales[isel] = np.where(take(mask) > 0, res["scales"], 0.0)
        if single.any():
            # "1-channel hack" (ppalign.py:196-201): fit_phase_shift of the one profile against its template
            # channel with the channel's noise, DM = the header's, nu_ref = the channel's frequency
            i1 = np.where(single)[0]
            ich = np.array([int(np.where(mask[i] > 0)[0][0]) for i in i1])
            r1 = eng.fit_phase_shift_batch(ports[i1, ich], model_port[ich], noise=errs[i1, ich], Ns=nbin,
                                           finish='simplex')
            phase[i1], DM[i1], nu_ref[i1] = r1[:, 0], DM_guess, f2[i1, ich]
            scales[i1, ich] = r1[:, 2]
        w_acc = np.where(mask > 0, scales / errs ** 2.0, 0.0)
        aligned, totw = eng.align_accumulate(ports, f2, Ps, phase, DM, nu_ref, w_acc)
        good = totw > 0
        aligned[good] /= totw[good, None]
        aligned[~good] = 0.0
        model_port = aligned
    if norm in ("mean", "max", "prof", "rms", "abs"):
        model_port = normalize_portrait(model_port, norm)
    return (model_port, res) if return_fits else model_port
