"""Host-side synthetic subints for tests, built with the ORACLE's helper
restatements; same draw order as tests/golden/make_golden.py:make_inputs so a
seed regenerates the arrays the reference was fed (to rounding; bitwise at 64x256)."""
import hashlib
import os

import numpy as np

from oracle import pptoas_oracle as orc

GMODEL = os.path.join(os.path.dirname(__file__), "golden", "example.gmodel")
P_EXAMPLE = 1.0 / 345.67890123456789


def input_sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def band(C, nu0=1500.0, bw=800.0):
    d = bw / C
    return np.linspace(nu0 - bw / 2 + d / 2, nu0 + bw / 2 - d / 2, C)


def model_portrait(C, B, nu0=1500.0, bw=800.0, P=P_EXAMPLE):
    freqs = band(C, nu0, bw)
    return freqs, orc.read_model_portrait(open(GMODEL).read(),
                                          orc.get_bin_centers(B), freqs, P)


def make_inputs(C, B, seed, DM0=0.0, sigma=0.05, scint=False, GM=None,
                tau_us=None, alpha=-4.0, nu0=1500.0, bw=800.0, model=None):
    rng = np.random.default_rng(seed)
    P = P_EXAMPLE
    freqs = band(C, nu0, bw)
    if model is None:
        _, model = model_portrait(C, B, nu0, bw, P)
    phi_inj = rng.uniform(-0.5, 0.5)
    dDM_inj = rng.normal(3e-4, 2e-4)
    GM_inj = 0.0 if GM is None else rng.normal(GM, 0.05)
    port = model.copy()
    if tau_us is not None:
        taus = orc.scattering_times(tau_us * 1e-6 / P, alpha, freqs, nu0)
        port = np.fft.irfft(orc.scattering_portrait_FT(taus, B) *
                            np.fft.rfft(port, axis=-1), axis=-1)
    port = orc.rotate_portrait_full(port, -phi_inj, -(DM0 + dDM_inj), -GM_inj,
                                    freqs, np.inf, np.inf, P)
    if scint:
        pars = []
        for _ in range(3):
            pars += [rng.uniform(0, 1.0), rng.chisquare(5.0), rng.uniform(0, 1)]
        port = orc.add_scintillation(port, params=pars)
    data = port + rng.normal(0.0, sigma, size=port.shape)
    return dict(data=data, model=model, freqs=freqs, errs=np.full(C, sigma), P=P,
                phi_inj=phi_inj, dDM_inj=dDM_inj, DM0=DM0, GM_inj=GM_inj)


def caller_guess(inp, fit_scat=False, log10_tau=True, tau_guess_rot=None,
                 alpha_guess=-4.0):
    """get_TOAs preamble (pptoas.py:399-460) with unit SNRs, oracle functions."""
    freqs, P = inp["freqs"], inp["P"]
    nu_mean = freqs.mean()
    nu_fit = orc.guess_fit_freq(freqs, None)
    DM_guess = inp["DM0"]
    rot_prof = orc.rotate_data(inp["data"], 0.0, DM_guess, P, freqs,
                               nu_mean).mean(axis=0)
    B = inp["data"].shape[1]
    tau_guess, a_guess = 0.0, 0.0
    mprof = inp["model"].mean(axis=0)
    if fit_scat:
        a_guess = alpha_guess
        tau_guess = 0.0 if tau_guess_rot is None else tau_guess_rot
        mprof = np.fft.irfft(orc.scattering_portrait_FT(
            np.array([tau_guess]), B)[0] * np.fft.rfft(mprof))
    fps = orc.fit_phase_shift(rot_prof, mprof, Ns=100)
    if fit_scat and log10_tau:
        if tau_guess == 0.0:
            tau_guess = B ** -1
        tau_guess = np.log10(tau_guess)
    phi_guess = orc.phase_transform(fps.phase, DM_guess, nu_mean, nu_fit, P,
                                    mod=True)
    return dict(nu_fit=nu_fit, init_params=np.array(
        [phi_guess, DM_guess, 0.0, tau_guess, a_guess]))


# ---- host restatement of the device generator (csrc/pp_extra.h k_synth) ----
def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return c0, c1, c2, c3


def philox_normal_pairs(seed, sub, chan, npair):
    """N(0,1) pairs for bins (2j, 2j+1), j < npair, of (subint, channel)."""
    j = np.arange(npair, dtype=np.uint64)
    ones = np.ones(npair, dtype=np.uint64)
    c = _philox4x32_10(j, ones * np.uint64(chan), ones * np.uint64(sub & 0xFFFFFFFF),
                       ones * np.uint64(sub >> 32), seed & 0xFFFFFFFF, seed >> 32)
    a = (c[0] << np.uint64(32)) | c[1]
    b = (c[2] << np.uint64(32)) | c[3]
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)


def device_recipe_host(model, freqs, P, inj, sigma, seed, first_subint):
    """data[i] = rotate(model, -phi, -DM, -GM; nu_ref = inf) + sigma N(0,1)."""
    C, B = model.shape
    out = np.empty((len(P), C, B))
    for i in range(len(P)):
        port = orc.rotate_portrait_full(model, -inj[i, 0], -inj[i, 1], -inj[i, 2],
                                        freqs, np.inf, np.inf, P[i])
        for n in range(C):
            z0, z1 = philox_normal_pairs(seed, first_subint + i, n, B // 2)
            port[n, 0::2] += sigma * z0
            port[n, 1::2] += sigma * z1
        out[i] = port
    return out
