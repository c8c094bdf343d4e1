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
