"""Gaussian-component model portraits (.gmodel files) on the host.

Input preparation for the fit, not part of the timed path: builds the
nchan x nbin template the engine is handed.  Same conventions as the
reference's read_model / gen_gaussian_portrait / gaussian_profile
(pplib.py:2867-2953, 853-930, 770-825, 996-1046), written channel-vectorised.
Scattered models (TAU != 0) are convolved in the Fourier domain
(pplib.py:915-922, 4049-4095).
"""
import numpy as np

from .pplib import get_bin_centers, scattering_alpha

# parameters of the reference's examples/example.gmodel (data, not code):
# three Gaussians, power-law evolution of loc / wid / amp about 1300 MHz
EXAMPLE_GMODEL = """MODEL   PSR_1234-5678
CODE    000
FREQ    1300.00000
DC      0.00889801 1
TAU     0.00000000 1
ALPHA  -4.000      0
COMP01  0.21925557 1  -0.00518501 1   0.04823579 1  -2.08031160 1    5.13274758 1   -1.65717015 1
COMP02  0.23409622 1  -0.00271530 1   0.01573809 1   1.61520300 1    9.46117549 1   -2.07617616 1
COMP03  0.25844309 1   0.00288377 1   0.02348129 1  -3.30015260 1    2.71065613 1   -0.90424701 1
"""
EXAMPLE_PERIOD = 1.0 / 345.67890123456789   # examples/example.par F0


def parse_gmodel(text):
    """-> dict(name, code, nu_ref, ngauss, params, fit_flags, alpha, fit_alpha);
    params = [DC, TAU[s], (loc, m_loc, wid, m_wid, amp, m_amp) * ngauss]."""
    out = dict(name=None, code='000', nu_ref=None, alpha=scattering_alpha,
               fit_alpha=0)
    params, flags, comps = [0.0, 0.0], [0, 0], []
    for line in text.splitlines():
        tok = line.split()
        if not tok or tok[0].startswith('#'):
            continue
        key = tok[0]
        if key == 'MODEL':
            out['name'] = tok[1]
        elif key == 'CODE':
            out['code'] = tok[1]
        elif key == 'FREQ':
            out['nu_ref'] = np.float64(tok[1])
        elif key == 'DC':
            params[0], flags[0] = np.float64(tok[1]), int(tok[2])
        elif key == 'TAU':
            params[1], flags[1] = np.float64(tok[1]), int(tok[2])
        elif key == 'ALPHA':
            out['alpha'], out['fit_alpha'] = np.float64(tok[1]), int(tok[2])
        elif key[:4] == 'COMP':
            comps.append(tok)
    for tok in comps:
        params += [np.float64(v) for v in tok[1:13:2]]
        flags += [int(v) for v in tok[2:13:2]]
    out.update(ngauss=len(comps), params=np.array(params),
               fit_flags=np.array(flags))
    return out


def read_gmodel(path):
    with open(path) as fh:
        return parse_gmodel(fh.read())


def _evolve(freqs, nu_ref, value, evol, code):
    if code == '0':   # power law, computed in logs like the reference
        return np.exp(np.outer(np.log(freqs) - np.log(nu_ref), evol) +
                      np.log(value)[None, :])
    return np.outer(freqs - nu_ref, evol) + value[None, :]


def gaussian_components(nbin, loc, wid):
    """Unit-peak Gaussians for arrays loc, wid of any (equal) shape ->
    shape + (nbin,).  FWHM wid <= 0 gives zeros; values beyond 20 sigma are
    zero; the peak is normalised with respect to `loc`, not `loc % 1`."""
    loc = np.asarray(loc, dtype=np.float64)[..., None]
    wid = np.asarray(wid, dtype=np.float64)[..., None]
    x = get_bin_centers(nbin)
    ok = wid > 0.0
    sigma = np.where(ok, wid, 1.0) / (2 * np.sqrt(2 * np.log(2)))
    mean = loc % 1.0
    xw = np.where(mean < 0.5, np.where(x > mean + 0.5, x - 1.0, x),
                  np.where(x < mean - 0.5, x + 1.0, x))
    z = (xw - mean) / sigma
    val = np.where(np.fabs(z) < 20.0, np.exp(-0.5 * z ** 2) /
                   (sigma * np.sqrt(2 * np.pi)), 0.0)
    ipk = val.argmax(axis=-1)[..., None]
    vpk = np.take_along_axis(val, ipk, axis=-1)
    zpk = (np.take_along_axis(xw, ipk, axis=-1) - loc) / sigma
    with np.errstate(divide='ignore', invalid='ignore'):
        fact = np.where(vpk > 0.0, np.exp(-0.5 * zpk ** 2) / vpk, 0.0)
    return np.where(ok, fact * val, 0.0)


def gaussian_portrait(model, freqs, nbin, P=None):
    """nchan x nbin portrait of a parsed .gmodel at the given frequencies."""
    freqs = np.asarray(freqs, dtype=np.float64)
    p, code, nu_ref = model['params'], model['code'], model['nu_ref']
    locs = _evolve(freqs, nu_ref, p[2::6], p[3::6], code[0])
    wids = _evolve(freqs, nu_ref, p[4::6], p[5::6], code[1])
    amps = _evolve(freqs, nu_ref, p[6::6], p[7::6], code[2])
    port = p[0] + (amps[..., None] * gaussian_components(nbin, locs, wids)).sum(1)
    if p[1] != 0.0:
        if P is None:
            raise ValueError("need the period P for a model with TAU != 0")
        taus = (p[1] / P) * (freqs / nu_ref) ** model['alpha']    # [rot]
        k = np.arange(nbin // 2 + 1)
        scat = 1.0 / (1.0 + 2j * np.pi * np.outer(taus, k))
        port = np.fft.irfft(scat * np.fft.rfft(port, axis=-1), axis=-1)
    return port


def example_model(nchan, nbin, nu0=1500.0, bw=800.0):
    """(freqs, portrait, P) of the reference's example pulsar on a band of
    nchan channels centred at nu0 (make_fake_pulsar's channelisation,
    pplib.py:3236-3240)."""
    d = bw / nchan
    freqs = np.linspace(nu0 - bw / 2 + d / 2, nu0 + bw / 2 - d / 2, nchan)
    model = parse_gmodel(EXAMPLE_GMODEL)
    return freqs, gaussian_portrait(model, freqs, nbin, EXAMPLE_PERIOD), \
        EXAMPLE_PERIOD
