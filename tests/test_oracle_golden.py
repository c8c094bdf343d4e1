"""Pin the CPU oracle (oracle/pptoas_oracle.py) against golden vectors made by
the true reference (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import pptoas_oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FPF = sorted(os.path.basename(p)[:-4] for p in
             glob.glob(os.path.join(GOLDEN, "fpf_64x256_*.npz")) +
             glob.glob(os.path.join(GOLDEN, "fpf_128x512_*.npz")) +
             # row lengths that are no power of two (tests/golden/make_golden_nbin.py)
             glob.glob(os.path.join(GOLDEN, "fpf_48x1000_*.npz")) +
             glob.glob(os.path.join(GOLDEN, "fpf_40x100_*.npz")) +
             glob.glob(os.path.join(GOLDEN, "fpf_24x1536_*.npz")))

# tolerances: north_star bar is 1e-9 (phase) / 1e-6 (DM); the oracle is held
# tighter where the reference itself is reproducible to rounding
PHI_TOL = {"default": 1e-12, "fpf_64x256_phiDMGM": 1e-9, "fpf_64x256_scat": 1e-9,
           "fpf_64x256_phiDMtau": 1e-9, "fpf_64x256_scat_lin": 1e-9,
           "fpf_64x256_all5": 1e-9}
DM_TOL = {"default": 1e-11, "fpf_64x256_phiDMGM": 1e-6, "fpf_64x256_scat": 1e-7,
          "fpf_64x256_phiDMtau": 1e-7, "fpf_64x256_scat_lin": 1e-7,
          "fpf_64x256_all5": 1e-6}


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _run(g, **kw):
    nu_outs = [None if np.isnan(v) else float(v) for v in g["nu_outs"]]
    if "option" in g.files:     # goldens of the non-default switches
        kw.setdefault("option", int(g["option"]))
        kw.setdefault("is_toa", bool(g["is_toa"]))
    return orc.fit_portrait_full(
        g["data"], g["model"], g["init_params"], float(g["P"]), g["freqs"],
        list(g["nu_fits"]), nu_outs, g["errs"], list(g["fit_flags"]),
        log10_tau=bool(g["log10_tau"]), **kw)


@pytest.mark.parametrize("name", FPF)
def test_fit_portrait_full_matches_reference(name):
    g = _load(name)
    r = _run(g)
    dphi = abs(r.phi - float(g["out_phi"]))
    dphi = min(dphi, abs(dphi - 1.0))
    assert dphi < PHI_TOL.get(name, PHI_TOL["default"]), dphi
    assert abs(r.DM - float(g["out_DM"])) < DM_TOL.get(name, DM_TOL["default"])
    loose = name in PHI_TOL
    rt = 1e-5 if loose else 1e-9
    np.testing.assert_allclose(r.params, g["out_params"], rtol=rt, atol=1e-9)
    np.testing.assert_allclose(r.param_errs, g["out_param_errs"], rtol=1e-6)
    np.testing.assert_allclose([r.nu_DM, r.nu_GM, r.nu_tau],
                               [g["out_nu_DM"], g["out_nu_GM"], g["out_nu_tau"]],
                               rtol=1e-6 if loose else 1e-10)
    np.testing.assert_allclose(r.scales, g["out_scales"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r.scale_errs, g["out_scale_errs"], rtol=1e-7)
    np.testing.assert_allclose(r.channel_snrs, g["out_channel_snrs"], rtol=1e-7,
                               atol=1e-8)
    cov = g["out_covariance_matrix"]
    cscale = np.sqrt(np.outer(np.diag(cov), np.diag(cov)))
    # off-diagonals are ~0 by construction at the zero-covariance frequency
    assert np.all(np.abs(r.covariance_matrix - cov) <= 1e-5 * np.abs(cov) +
                  1e-8 * cscale)
    np.testing.assert_allclose(r.chi2, g["out_chi2"], rtol=1e-11)
    np.testing.assert_allclose(r.red_chi2, g["out_red_chi2"], rtol=1e-11)
    np.testing.assert_allclose(r.snr, g["out_snr"], rtol=1e-9)


@pytest.mark.parametrize("name", FPF)
def test_objective_points(name):
    g = _load(name)
    B = g["data"].shape[1]
    dFT = np.fft.rfft(g["data"], axis=-1)
    dFT[:, 0] = 0
    mFT = np.fft.rfft(g["model"], axis=-1)
    mFT[:, 0] = 0
    errs_FT = g["errs"] * np.sqrt(B / 2.0)
    args = (dFT, mFT, errs_FT, float(g["P"]), g["freqs"], g["nu_fits"][0],
            g["nu_fits"][1], g["nu_fits"][2], list(g["fit_flags"]),
            bool(g["log10_tau"]))
    for i, p in enumerate(g["obj_points"]):
        f = orc.fit_portrait_full_function(p, *args)
        gr = orc.fit_portrait_full_function_deriv(p, *args)
        hs = orc.fit_portrait_full_function_2deriv(p, *args)
        np.testing.assert_allclose(f, g["obj_f"][i], rtol=1e-13)
        scale = np.abs(g["obj_grad"][i]).max() + 1e-300
        np.testing.assert_allclose(gr, g["obj_grad"][i], rtol=1e-9,
                                   atol=1e-12 * scale)
        hscale = np.abs(g["obj_hess"][i]).max()
        np.testing.assert_allclose(hs, g["obj_hess"][i], rtol=1e-9,
                                   atol=1e-12 * hscale)


def test_scalars_512x1024_regenerated_inputs():
    """Larger shape: inputs are re-made by the oracle's own helpers from the
    seed (checked against summaries of what the reference was fed), outputs compared."""
    from tests.synth_host import make_inputs
    g = _load("fpf_512x1024_phiDM_scalars")
    inp = make_inputs(int(g["C"]), int(g["B"]), int(g["seed"]))
    np.testing.assert_allclose(inp["data"][::16, ::16], g["data_sample"],
                               rtol=0, atol=1e-13)
    np.testing.assert_allclose(inp["model"][::16, ::16], g["model_sample"],
                               rtol=0, atol=1e-13)
    np.testing.assert_allclose(inp["data"].sum(1), g["data_rowsum"], atol=1e-10)
    np.testing.assert_allclose(inp["data"].sum(0), g["data_colsum"], atol=1e-10)
    r = orc.fit_portrait_full(inp["data"], inp["model"], g["init_params"],
                              inp["P"], inp["freqs"], list(g["nu_fits"]),
                              [None] * 3, inp["errs"], [1, 1, 0, 0, 0],
                              log10_tau=False)
    assert abs(r.phi - float(g["out_phi"])) < 1e-12
    assert abs(r.DM - float(g["out_DM"])) < 1e-11
    np.testing.assert_allclose(r.param_errs, g["out_param_errs"], rtol=1e-7)
    np.testing.assert_allclose(r.scale_errs, g["out_scale_errs"], rtol=1e-7)
    np.testing.assert_allclose(r.nu_DM, g["out_nu_DM"], rtol=1e-11)
    np.testing.assert_allclose(r.red_chi2, g["out_red_chi2"], rtol=1e-11)


def test_fit_phase_shift_rows():
    g = _load("fit_phase_shift_256")
    for row in g["rows"]:
        shift, noise = row[0], (None if np.isnan(row[1]) else row[1])
        d = orc.rotate_data(g["prof"], -shift)
        r = orc.fit_phase_shift(d, g["model_prof"], noise=noise, Ns=100)
        assert abs(r.phase - row[2]) < 1e-9          # same simplex path
        np.testing.assert_allclose(
            [r.phase_err, r.scale, r.scale_err, r.snr, r.red_chi2], row[3:],
            rtol=1e-7)


def test_legacy_fit_portrait():
    g = _load("legacy_fit_portrait_64x256")
    r = orc.fit_portrait(g["data"], g["model"], g["init_params"], float(g["P"]),
                         g["freqs"], float(g["nu_fit"]), None, g["errs"])
    assert abs(r.phase - float(g["out_phase"])) < 1e-9
    assert abs(r.DM - float(g["out_DM"])) < 1e-8
    np.testing.assert_allclose(r.scales, g["out_scales"], rtol=1e-8)
    np.testing.assert_allclose(r.scale_errs, g["out_scale_errs"], rtol=1e-12)
    np.testing.assert_allclose(r.nu_ref, g["out_nu_ref"], rtol=1e-9)
    np.testing.assert_allclose([r.phase_err, r.DM_err, r.snr, r.red_chi2],
                               [g["out_phase_err"], g["out_DM_err"], g["out_snr"],
                                g["out_red_chi2"]], rtol=1e-7)


def test_helper_restatements():
    g = _load("helpers_64x256")
    text = open(os.path.join(GOLDEN, "example.gmodel")).read()
    P = 1.0 / 345.67890123456789
    np.testing.assert_array_equal(orc.get_bin_centers(256), g["phases"])
    model = orc.read_model_portrait(text, g["phases"], g["freqs"], P)
    np.testing.assert_allclose(model, g["model"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(orc.gaussian_profile(256, 0.9961, 0.031), g["gp"],
                               rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(orc.gaussian_profile(256, 1.23, 0.11), g["gp2"],
                               rtol=1e-14, atol=1e-300)
    assert orc.guess_fit_freq(g["freqs"]) == float(g["nu_fit"])
    np.testing.assert_allclose(
        orc.guess_fit_freq(g["freqs"], np.linspace(1.0, 3.0, 64)),
        g["nu_fit_snr"], rtol=1e-15)
    np.testing.assert_allclose(
        orc.phase_transform(0.3, 34.5, 1500.0, 1200.0, P, mod=True),
        g["phase_tr"], rtol=1e-14)


def test_oracle_instrumental_response_matches_reference():
    """instrumental_response_port_FT / gaussian_profile_FT of the oracle against the
    arrays the true reference produced for the gettoas_ird golden (pptoaslib.py:14-50,
    145-179; tests/golden/make_golden_gettoas.py)."""
    import os
    from oracle import pptoas_oracle as orc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gettoas_ird.npz"))
    ok = np.where(g["weights"][0] > 0)[0]
    resp = orc.instrumental_response_port_FT(
        g["subints"].shape[-1], g["freqs"][0][ok], float(g["out_ird_DM"]), float(g["Ps"][0]),
        [float(v) for v in g["out_ird_wids"]], [str(v) for v in g["out_ird_types"]])
    np.testing.assert_allclose(resp, g["out_ird_resp"], rtol=1e-13, atol=1e-15)
    gft = orc.gaussian_profile_FT(g["subints"].shape[-1], 0.3, 0.02, 1.7)
    np.testing.assert_allclose(gft, g["out_ird_gauss_FT"], rtol=1e-13, atol=1e-13)
