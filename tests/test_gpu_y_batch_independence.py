"""A subint's answer is a function of that subint alone -- at the stated shapes.

The reference fits subints in a plain loop with no cross-iteration state
(/root/reference pptoas.py:344-489): what it returns for one subint cannot depend on which other
subints are in the archive.  Here a batch shares kernel launches, so every reduction over channels
must add its terms in an order fixed by the BAND (never by the batch size): the channel chunks of the
evaluators / seed / moment kernels (`chunking` in csrc/pp_toas.hip, PP_CHUNK_CHANNELS) and the channel
runs of pp_reference_phase_seed (csrc/pp_extra_api.h) are functions of nchan only.  Until round 4 both
grew with 1 / nsub, and at nchan >= 128 the rounding of f -- and through SciPy's 1-ulp exit tests up
to ~1e-9 rot of an answer -- depended on the neighbours.

The same subints are fitted alone, in a batch of 7 and in a batch of 512, at configs[3]'s shape
(2048 x 2048, scattering, both solvers) and at the headline shape (4096 x 2048: device seed, the
reference's seed inside the fit's pass, pp_reference_phase_seed, a fallback list of poor guesses, GM);
EVERY output must be bit-identical."""
import argparse

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NBIG = 512
OUT_KEYS = ("params", "param_errs", "nu_refs", "cov", "chi2", "red_chi2", "snr", "nfeval", "return_code",
            "npass", "scales", "scale_errs", "channel_snrs")


def _args(**kw):
    d = dict(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05, truth_guesses=False,
             measured_noise=False, method="trust-ncg")
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.fixture(scope="module")
def eng():
    from pulseportraiture_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _fit(eng, b, lo, hi, x0=None, **kw):
    x0 = b.x0 if x0 is None else x0
    n = hi - lo
    call = dict(errs=b.errs_dev[lo:hi], nu_fits=np.full((n, 3), b.nu_fit), fit_flags=b.flags,
                log10_tau=b.log10_tau, per_channel=True)
    call.update(kw)
    return eng.fit_batch(b.data[lo:hi], b.freqs, b.P[lo:hi], x0[lo:hi], **call)


def _same(whole, part, lo, hi, what):
    for k in OUT_KEYS:
        np.testing.assert_array_equal(np.asarray(whole[k])[lo:hi], np.asarray(part[k]), err_msg="%s: %s" % (what, k))


def _alone_in_7_in_512(eng, b, what, x0=None, picks=(3, 200, NBIG - 1), **kw):
    whole = _fit(eng, b, 0, NBIG, x0=x0, **kw)
    seven = _fit(eng, b, 0, 7, x0=x0, **kw)
    _same(whole, seven, 0, 7, what + " [7 of 512]")
    for i in picks:
        one = _fit(eng, b, i, i + 1, x0=x0, **kw)
        _same(whole, one, i, i + 1, what + " [subint %d alone]" % i)
    # a batch that starts somewhere else in the job and has another length
    mid = _fit(eng, b, 100, 161, x0=x0, **kw)
    _same(whole, mid, 100, 161, what + " [61 from the middle]")
    return whole


@pytest.mark.timeout(900)
@pytest.mark.parametrize("method", ["trust-ncg", "newton"])
def test_scattering_fits_at_2048x2048_do_not_depend_on_the_batch(eng, method):
    """configs[3]: the evaluation loop over the stored cross-spectrum (k_eval_scat: channel chunks),
    the first evaluation inside the transform, the model pass and its solve; the Newton solver's
    coarse iteration on every 16th channel."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    b = bench.Batch(eng, _args(), dev, "cfg4-2048x2048-scat", NBIG, "f64", 0)
    try:
        r = _alone_in_7_in_512(eng, b, "2048x2048 scattering " + method, method=method)
        assert (r["return_code"] == 2).all() or method == "newton"
        assert (r["nfeval"] > 3).all()
    finally:
        b.free()


@pytest.mark.timeout(900)
def test_headline_shape_flows_do_not_depend_on_the_batch(eng):
    """4096 x 2048: the one-pass flow (independent by construction: one wave per row), the device
    phase seed (pilot pass + k_seed_accum's channel chunks), the reference's seed formed inside the
    fit's pass, pp_reference_phase_seed's channel runs, a GM fit, and a batch in which a third of the
    subints have poor DM / phase guesses: re-expansions, the list of subints transformed again with
    the cross-spectrum stored, evaluations over it (k_eval_fast: channel chunks)."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    b = bench.Batch(eng, _args(), dev, "toa-4096x2048-phiDM", NBIG, "f64", 0)
    try:
        # pp_reference_phase_seed (the two-pass route of get_TOAs' default flow): 512 / 7 / 1
        w = np.ones((NBIG, b.C))
        nu_mean = float(b.freqs.mean())
        kw = dict(DM=np.full(NBIG, 34.56789), nu_DM=nu_mean, Ns=100, finish='simplex')

        def seed(lo, hi):
            k = dict(kw, DM=kw["DM"][lo:hi])
            return eng.reference_phase_seed(b.data[lo:hi], b.freqs, b.P[lo:hi], w[lo:hi], b.seed_prof, **k)[:, :6]
        s_all = seed(0, NBIG)
        np.testing.assert_array_equal(s_all[:7], seed(0, 7))
        for i in (3, 200, NBIG - 1):
            np.testing.assert_array_equal(s_all[i:i + 1], seed(i, i + 1))
        # the plain one-pass fit from the caller's guesses
        for method in ("trust-ncg", "newton"):
            _alone_in_7_in_512(eng, b, "4096x2048 one-pass " + method, method=method)
        # device phase seed inside the fit
        _alone_in_7_in_512(eng, b, "4096x2048 device seed", seed_ns=100)
        # the reference's own guess formed inside the fit's single pass
        rs = dict(weights=None, model_profs=b.seed_prof, nu_mean=b.nu_mean, Ns=100, finish='simplex')

        def rs_of(lo, hi):
            return dict(rs, nu_mean=b.nu_mean[lo:hi])
        whole = _fit(eng, b, 0, NBIG, ref_seed=rs_of(0, NBIG))
        for lo, hi in ((0, 7), (3, 4), (200, 201), (100, 161)):
            part = _fit(eng, b, lo, hi, ref_seed=rs_of(lo, hi))
            _same(whole, part, lo, hi, "4096x2048 reference seed in the pass [%d, %d)" % (lo, hi))
            np.testing.assert_array_equal(whole["seed_phase"][lo:hi], part["seed_phase"])
        # poor guesses: re-expansion, fallback list, evaluations over the stored cross-spectrum
        rng = np.random.default_rng(77)
        x0 = b.x0.copy()
        u = rng.random(NBIG)
        poor_dm = u < 0.2
        poor_phi = (u >= 0.2) & (u < 0.33)
        x0[poor_dm, 1] += rng.choice([-1, 1], poor_dm.sum()) * rng.uniform(2e-3, 1e-2, poor_dm.sum())
        x0[poor_phi, 0] = (x0[poor_phi, 0] + rng.choice([-1, 1], poor_phi.sum()) *
                           rng.uniform(0.02, 0.06, poor_phi.sum()) + 0.5) % 1.0 - 0.5
        x0[3, 0] = (x0[3, 0] + 0.04 + 0.5) % 1.0 - 0.5          # (the subints fitted alone are poor ones too)
        x0[200, 1] += 6e-3
        for method in ("trust-ncg", "newton"):
            r = _alone_in_7_in_512(eng, b, "4096x2048 poor guesses " + method, x0=x0, method=method)
            assert (r["npass"] > 1).sum() >= 50 and (r["npass"] == 1).sum() >= 50, np.bincount(r["npass"])
    finally:
        b.free()
    b = bench.Batch(eng, _args(), dev, "cfg3-4096x2048-phiDMGM", NBIG, "f64", 0)
    try:
        _alone_in_7_in_512(eng, b, "4096x2048 GM")
    finally:
        b.free()


@pytest.mark.timeout(600)
def test_wide_band_small_shapes_do_not_depend_on_the_batch(eng):
    """The same property where the old rule changed its chunking most often: bands of 128 ... 1024
    channels, 13 subints against each subint alone, scattering and the forced evaluation loop
    (taylor = 0: k_eval_fast for every subint), masks included."""
    from tests.synth_host import make_inputs, caller_guess, model_portrait
    rng = np.random.default_rng(5)
    for C, B, scat in ((128, 256, True), (320, 512, True), (1024, 256, False), (576, 128, False), (200, 1000, True)):
        freqs, model = model_portrait(C, B)
        eng.set_model(model)
        N = 13
        data, x0, nuf, masks = [], [], [], []
        for i in range(N):
            inp = make_inputs(C, B, 8800 + i, model=model, tau_us=(25.0 if scat else None), sigma=0.05)
            g = caller_guess(inp, fit_scat=scat, log10_tau=scat,
                             tau_guess_rot=(1.3 * 25e-6 / inp["P"]) if scat else None)
            data.append(inp["data"]); x0.append(g["init_params"]); nuf.append([g["nu_fit"]] * 3)
            masks.append((rng.random(C) > 0.1).astype(np.uint8))
        data, x0, nuf, masks = map(np.array, (data, x0, nuf, masks))
        P = np.full(N, inp["P"])
        flags = [1, 1, 0, 1, 1] if scat else [1, 1, 0, 0, 0]
        for method in ("trust-ncg", "newton"):
            for taylor in ((1,) if scat else (1, 0)):
                eng.set_option("taylor", taylor)
                try:
                    kw = dict(errs=np.full((N, C), 0.05), chan_mask=masks, nu_fits=nuf, fit_flags=flags,
                              log10_tau=scat, method=method)
                    whole = eng.fit_batch(data, freqs, P, x0, **kw)
                    for i in range(N):
                        k1 = dict(kw, errs=kw["errs"][i:i + 1], chan_mask=masks[i:i + 1], nu_fits=nuf[i:i + 1])
                        one = eng.fit_batch(data[i:i + 1], freqs, P[i:i + 1], x0[i:i + 1], **k1)
                        _same(whole, one, i, i + 1, "%dx%d scat=%s %s taylor=%d subint %d" % (C, B, scat, method, taylor, i))
                finally:
                    eng.set_option("taylor", 1)
