"""Spline (PCA + B-spline) model portraits on the host: `read_spline_model` and
`gen_spline_portrait` with the conventions of pplib.py:932-956, 2955-2987.

A `.spl` file is the pickle written by ppspline.write_model:
[model_name, source, datafile, mean_prof, eigvec, tck].  Input preparation for
the fit, outside the timed path."""
import pickle

import numpy as np
import scipy.interpolate as si
import scipy.signal as ss


def gen_spline_portrait(mean_prof, freqs, eigvec, tck, nbin=None):
    """Model portrait at `freqs`: mean profile + (spline-evaluated PCA
    coordinates) x eigenvectors; resampled in phase if nbin differs."""
    mean_prof = np.asarray(mean_prof, dtype=np.float64)
    freqs = np.asarray(freqs, dtype=np.float64)
    if not eigvec.shape[1]:
        port = np.tile(mean_prof, len(freqs)).reshape(len(freqs), len(mean_prof))
    else:
        proj_port = np.array(si.splev(freqs, tck, der=0, ext=0)).T
        port = np.dot(proj_port, eigvec.T) + mean_prof
    if nbin is not None and len(mean_prof) != nbin:
        shift = 0.5 * (nbin ** -1 - len(mean_prof) ** -1)
        port = ss.resample(port, nbin, axis=1)
        # ss.resample introduces a phase shift; rotate it out (rotate_portrait)
        pFT = np.fft.rfft(port, axis=1)
        pFT *= np.exp(np.arange(pFT.shape[1]) * 2.0j * np.pi * shift)
        port = np.fft.irfft(pFT)
    return port


def read_spline_model(modelfile, freqs=None, nbin=None, quiet=False):
    """(name, source, datafile, mean_prof, eigvec, tck) of a .spl file, or
    (name, portrait) when freqs is given -- like the reference."""
    if not quiet:
        print("Reading model from %s..." % modelfile)
    with open(modelfile, 'rb') as fh:
        try:
            content = pickle.load(fh)
        except UnicodeDecodeError:       # pickles written by Python 2
            fh.seek(0)
            content = pickle.load(fh, encoding='latin1')
    modelname, source, datafile, mean_prof, eigvec, tck = content
    if freqs is None:
        return (modelname, source, datafile, mean_prof, eigvec, tck)
    return (modelname, gen_spline_portrait(mean_prof, freqs, eigvec, tck, nbin))


def spline_device_args(mean_prof, eigvec, tck, nbin=None):
    """What the device synthesis needs (Engine.set_model_spline): the basis
    [ncomp + 1, nbin] = mean profile and eigenvectors -- resampled to nbin exactly
    as gen_spline_portrait resamples the finished portrait (the operation is linear
    along phase, so it commutes with the sum over components) --, the knots, the
    [ncomp, nknots] coefficients and the degree of `tck`."""
    mean_prof = np.asarray(mean_prof, dtype=np.float64)
    eigvec = np.asarray(eigvec, dtype=np.float64).reshape(len(mean_prof), -1)
    basis = np.vstack([mean_prof[None, :], eigvec.T])
    if nbin is not None and len(mean_prof) != nbin:
        shift = 0.5 * (nbin ** -1 - len(mean_prof) ** -1)
        basis = ss.resample(basis, nbin, axis=1)
        bFT = np.fft.rfft(basis, axis=1)
        bFT *= np.exp(np.arange(bFT.shape[1]) * 2.0j * np.pi * shift)
        basis = np.fft.irfft(bFT)
    ncomp = eigvec.shape[1]
    if ncomp:
        t = np.ascontiguousarray(tck[0], dtype=np.float64)
        coefs = np.ascontiguousarray(np.array([np.asarray(cj, dtype=np.float64) for cj in tck[1]]))
        if coefs.shape[1] < len(t):       # scipy >= 1.? may trim the unused tail
            coefs = np.hstack([coefs, np.zeros((ncomp, len(t) - coefs.shape[1]))])
        k = int(tck[2])
    else:
        t, coefs, k = np.zeros(1), np.zeros((1, 1)), 3
    return np.ascontiguousarray(basis), t, np.ascontiguousarray(coefs[:, :len(t)]), k
