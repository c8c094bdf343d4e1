"""Reference-shaped entry points of pplib that sit on the wideband-TOA path.

Mirrors (same names, argument meaning, result fields) of the reference's
pplib.py for: module settings (:45-83), RCSTRINGS (:111-119), DataBunch
(:125-136), fit_phase_shift (:2054-2100), the legacy fit_portrait (:2102-2204),
and the small host-side helpers the callers use between fits
(get_bin_centers :671-684, DM_delay :2577-2590, phase_transform :2592-2616,
guess_fit_freq :2618-2632).  All array arithmetic of the fits themselves runs
in HIP kernels through pulseportraiture_amd.engine.
"""
import sys

import numpy as np

from .engine import default_engine

# ---- settings (pplib.py:45-83); Dconst must be bit-identical ---------------
Dconst_exact = 4.148808e3
Dconst_trad = 0.000241 ** -1
Dconst = Dconst_trad
scattering_alpha = -4.0
use_get_noise = True
default_noise_method = 'PS'
F0_fact = 0
wid_max = 0.25
default_model = '000'
binshift = 1.0

# return-code strings (pplib.py:111-119); the device solver reports
# 0 gradient-converged, 1 iteration limit, 2 stalled at rounding (the
# reference's usual trust-ncg exit), 3 NaN/singular
RCSTRINGS = {'-1': 'INFEASIBLE: Infeasible (low > up).',
             '0': 'LOCALMINIMUM: Local minima reach (|pg| ~= 0).',
             '1': 'FCONVERGED: Converged (|f_n-f_(n-1)| ~= 0.)',
             '2': 'XCONVERGED: Converged (|x_n-x_(n-1)| ~= 0.)',
             '3': 'MAXFUN: Max. number of function evaluations reach.',
             '4': 'LSFAIL: Linear search failed.',
             '5': 'CONSTANT: All lower bounds are equal to the upper bounds.',
             '6': 'NOPROGRESS: Unable to progress.',
             '7': 'USERABORT: User requested end of minimization.'}


class DataBunch(dict):
    """dict whose attributes are its keys; results stay mutable attribute-dicts
    because callers modify them after the fit (pplib.py:125-136)."""

    def __init__(self, **kwds):
        dict.__init__(self, kwds)
        self.__dict__ = self


class DataPortrait(object):
    """Holder of the data a model is fit to, built FROM ARRAYS.

    The reference's DataPortrait (pplib.py:138-650) loads PSRCHIVE archives and
    carries joining / normalising / plotting methods; only its single-archive
    field layout (pplib.py:306-327) is part of the fit path and is what this
    class provides: every key of the load_data DataBunch as an attribute, plus
    port, portx, freqsxs, noise_stdsxs, SNRsxs.  `data` is a DataBunch with the
    load_data fields (see pulseportraiture_amd.pptoas.data_from_arrays)."""

    def __init__(self, data=None, joinfile=None, quiet=False, **kwargs):
        if data is None or not isinstance(data, dict):
            raise RuntimeError("DataPortrait needs a DataBunch built from arrays "
                               "(PSRFITS loading requires PSRCHIVE)")
        self.init_params = []
        self.joinfile = joinfile
        self.njoin = 0
        self.join_params = []
        self.join_ichans = []
        self.all_join_params = []
        self.datafile = data.get("filename", "arrays")
        self.datafiles = [self.datafile]
        self.data = data
        for key in data.keys():
            setattr(self, key, data[key])
        if getattr(self, "source", None) is None:
            self.source = "noname"
        self.port = (self.masks * self.subints)[0, 0]
        self.portx = self.port[self.ok_ichans[0]]
        self.freqsxs = [self.freqs[0, self.ok_ichans[0]]]
        if self.noise_stds is not None:
            self.noise_stdsxs = np.asarray(self.noise_stds)[0, 0, self.ok_ichans[0]]
        self.SNRsxs = np.asarray(self.SNRs)[0, 0, self.ok_ichans[0]]
        if data.get("flux_prof") is not None:
            self.flux_profx = data["flux_prof"][self.ok_ichans[0]]


# ---- small host helpers ------------------------------------------------------
def get_bin_centers(nbin, lo=0.0, hi=1.0):
    lo, hi = np.double(lo), np.double(hi)
    diff = hi - lo
    return np.double(np.linspace(lo + diff / (nbin * 2), hi - diff / (nbin * 2),
                                 nbin))


def DM_delay(DM, freq, freq_ref=np.inf, P=None):
    delay = Dconst * DM * ((freq ** -2.0) - (freq_ref ** -2.0))
    return delay / P if P else delay


def phase_transform(phi, DM, nu_ref1=np.inf, nu_ref2=np.inf, P=None, mod=False):
    if P is None:
        P, mod = 1.0, False
    phi_prime = phi + (Dconst * DM * P ** -1 * (nu_ref2 ** -2.0 - nu_ref1 ** -2.0))
    if mod:
        phi_prime = np.where(abs(phi_prime) >= 0.5, phi_prime % 1, phi_prime)
        phi_prime = np.where(phi_prime >= 0.5, phi_prime - 1.0, phi_prime)
        if not phi_prime.shape:
            phi_prime = np.float64(phi_prime)
    return phi_prime


def guess_fit_freq(freqs, SNRs=None):
    freqs = np.asarray(freqs, dtype=np.float64)
    nu0 = (freqs.min() + freqs.max()) * 0.5
    if SNRs is None:
        SNRs = np.ones(len(freqs))
    diff = np.sum((freqs - nu0) * SNRs * freqs ** -2) / np.sum(SNRs * freqs ** -2)
    return nu0 + diff


# ---- Fourier rotation (device) ---------------------------------------------------
def rotate_data(data, phase=0.0, DM=0.0, Ps=None, freqs=None, nu_ref=np.inf):
    """Rotate and/or dedisperse a profile [nbin], portrait [nchan,nbin] or
    subint cube [nsub,npol,nchan,nbin]; positive phase / DM rotate to earlier
    phase (pplib.py:2338-2426).  Runs on the GPU."""
    data = np.asarray(data, dtype=np.float64)
    shape = data.shape
    if data.ndim == 1:
        cube = data[None, None]
    elif data.ndim == 2:
        cube = data[None]
    elif data.ndim == 4:
        cube = data.reshape(shape[0], shape[1] * shape[2], shape[3])
    else:
        print("Wrong number of dimensions.")
        return 0
    nsub, nch = cube.shape[0], cube.shape[1]
    if DM == 0.0 or freqs is None:
        fr, P, DM = np.full(nch, np.inf), np.ones(nsub), 0.0
    else:
        P = np.ones(nsub) * Ps
        fr = np.asarray(freqs, dtype=np.float64)
        if fr.ndim == 0:
            fr = np.full(shape[-2] if data.ndim > 1 else 1, float(fr))
        if data.ndim == 4:
            fr = np.broadcast_to(fr, (nsub, shape[2]))[:, None, :].repeat(shape[1], axis=1)
            fr = fr.reshape(nsub, nch)
    with np.errstate(divide='ignore'):
        out = default_engine().rotate_portraits(cube, fr, P, phi=phase, DM=DM, nu_DM=nu_ref)
    return out.reshape(shape)


def rotate_portrait(port, phase=0.0, DM=None, P=None, freqs=None, nu_ref=np.inf):
    """pplib.py:2428-2460."""
    if DM is None and freqs is None:
        return rotate_data(port, phase)
    return rotate_data(port, phase, DM, P, freqs, nu_ref)


def rotate_profile(profile, phase=0.0):
    return rotate_data(profile, phase)


# ---- 1-D FFTFIT ----------------------------------------------------------------
def weighted_mean(data, errs=1.0):
    """Weighted mean and its standard error, weights errs**-2 over the entries
    with errs > 0 (pplib.py:696-709)."""
    data = np.asarray(data, dtype=np.float64)
    if hasattr(errs, 'is_integer'):
        errs = np.ones(len(data))
    errs = np.asarray(errs, dtype=np.float64)
    ii = np.where(errs > 0.0)[0]
    w = errs[ii] ** -2.0
    return (data[ii] * w).sum() / w.sum(), w.sum() ** -0.5


def get_noise(data, method=default_noise_method, frac=4, chans=False):
    """Off-pulse noise from the mean of the top 1/frac of the power spectrum
    (get_noise / get_noise_PS, pplib.py:2206-2253; only the "PS" method).  Host
    helper for normalisation; inside a fit the engine measures the same quantity
    itself when errs is None."""
    if method != "PS":
        print("Unknown get_noise method.")
        return 0
    a = np.asarray(data, dtype=np.float64)
    rows = a if chans else a.reshape(1, -1)
    power = np.abs(np.fft.rfft(rows, axis=-1)) ** 2.0 / rows.shape[-1]
    kc = int((1 - frac ** -1) * power.shape[-1])
    noise = np.sqrt(power[:, kc:].mean(axis=-1))
    return noise if chans else noise[0]


def fit_phase_shift(data, model, noise=None, bounds=[-0.5, 0.5], Ns=100, finish='simplex'):
    """Fit a phase shift between a data and a model profile on the GPU (pplib.py:2054):
    Ns-point brute grid over `bounds` (both ends included), then -- like the
    reference, whose scipy.optimize.brute finishes with fmin -- the Nelder-Mead
    simplex to xtol = ftol = 1e-4, retraced step for step, so the returned phase is the
    reference's (~1e-5 rot from the optimum of the correlation).  finish='newton'
    returns the exact local optimum instead."""
    eng = default_engine()
    out = eng.fit_phase_shift_batch(np.asarray(data, dtype=np.float64)[None],
                                    np.asarray(model, dtype=np.float64)[None],
                                    noise=None if noise is None else [noise],
                                    bounds=bounds, Ns=Ns, finish=finish)[0]
    return DataBunch(phase=out[0], phase_err=out[1], scale=out[2],
                     scale_err=out[3], snr=out[4], red_chi2=out[5],
                     duration=out[6])


# ---- legacy 2-parameter fit ------------------------------------------------------
def fit_portrait(data, model, init_params, P, freqs, nu_fit=None, nu_out=None,
                 errs=None, bounds=[(None, None), (None, None)], id=None,
                 quiet=True):
    """(phase, DM) fit with the legacy result fields; runs the same device
    engine as fit_portrait_full with fit_flags = [1,1,0,0,0] (the reference's
    TNC minimiser is not reproduced: both converge to the same optimum)."""
    from .pptoaslib import fit_portrait_full
    x0 = [init_params[0], init_params[1], 0.0, 0.0, 0.0]
    r = fit_portrait_full(data, model, x0, P, freqs, [nu_fit] * 3, [nu_out] * 3,
                          errs, [1, 1, 0, 0, 0], log10_tau=False, sub_id=id, method='TNC',
                          is_toa=True, quiet=quiet)
    with np.errstate(divide='ignore', invalid='ignore'):
        scale_errs = np.abs(r.scales / r.channel_snrs)   # (p_n/sigma_n^2)^-1/2
    return DataBunch(phase=r.phi, phase_err=r.phi_err, DM=r.DM, DM_err=r.DM_err,
                     scales=r.scales, scale_errs=scale_errs, nu_ref=r.nu_DM,
                     covariance=r.covariance_matrix[0, 1], chi2=r.chi2,
                     red_chi2=r.red_chi2, snr=r.snr, duration=r.duration,
                     nfeval=r.nfeval, return_code=r.return_code)
