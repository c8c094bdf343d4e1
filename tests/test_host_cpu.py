"""Host-side logic of the product that needs no GPU: .gmodel portraits, the
result container, helper formulas, and the generator's RNG known answers."""
import os

import numpy as np

from pulseportraiture_amd import gmodel, pplib

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_gmodel_portrait_matches_reference():
    g = np.load(os.path.join(GOLDEN, "helpers_64x256.npz"))
    m = gmodel.gaussian_portrait(gmodel.parse_gmodel(gmodel.EXAMPLE_GMODEL),
                                 g["freqs"], 256, gmodel.EXAMPLE_PERIOD)
    np.testing.assert_allclose(m, g["model"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(gmodel.gaussian_components(256, 0.9961, 0.031),
                               g["gp"], rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(gmodel.gaussian_components(256, 1.23, 0.11),
                               g["gp2"], rtol=1e-14, atol=1e-300)
    file_model = gmodel.read_gmodel(os.path.join(GOLDEN, "example.gmodel"))
    np.testing.assert_array_equal(file_model["params"],
                                  gmodel.parse_gmodel(gmodel.EXAMPLE_GMODEL)["params"])
    assert np.all(gmodel.gaussian_components(64, 0.3, 0.0) == 0.0)


def test_helpers_match_reference():
    g = np.load(os.path.join(GOLDEN, "helpers_64x256.npz"))
    P = gmodel.EXAMPLE_PERIOD
    assert pplib.Dconst == 0.000241 ** -1 == 4149.377593360996
    np.testing.assert_array_equal(pplib.get_bin_centers(256), g["phases"])
    assert pplib.guess_fit_freq(g["freqs"]) == float(g["nu_fit"])
    np.testing.assert_allclose(pplib.guess_fit_freq(g["freqs"], np.linspace(1, 3, 64)),
                               g["nu_fit_snr"], rtol=1e-15)
    np.testing.assert_allclose(pplib.phase_transform(0.3, 34.5, 1500.0, 1200.0, P, True),
                               g["phase_tr"], rtol=1e-14)


def test_databunch_is_a_mutable_attribute_dict():
    r = pplib.DataBunch(phi=0.1, DM=3.0)
    r.TOA = 5
    r.DM *= 2
    assert r["TOA"] == 5 and r["DM"] == 6.0 and set(r) == {"phi", "DM", "TOA"}


def test_philox_known_answers():
    """Random123 known-answer vectors for Philox4x32-10 (the device generator's
    RNG, restated on the host in tests/synth_host.py)."""
    from tests.synth_host import _philox4x32_10, philox_normal_pairs
    c = _philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(v[0]) for v in c] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    m = 0xffffffff
    c = _philox4x32_10([m], [m], [m], [m], m, m)
    assert [int(v[0]) for v in c] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    z0, z1 = philox_normal_pairs(20260101, 7, 3, 50000)
    assert abs(z0.mean()) < 0.02 and abs(z0.std() - 1) < 0.02
    assert abs(z1.mean()) < 0.02 and abs(np.corrcoef(z0, z1)[0, 1]) < 0.02


def test_mjd_and_tim_line():
    from pulseportraiture_amd.pptoas import MJD, TOA, toa_string
    t = MJD(55000, 0.999999999) + MJD(0, 2e-9)
    assert t.intday() == 55001 and abs(t.fracday() - 1e-9) < 1e-15
    assert MJD(55000.25).intday() == 55000 and MJD(55000.25).fracday() == 0.25
    toa = TOA("a.fits", np.inf, MJD(55000, 0.123456789012345), 0.1234, "GBT", "1",
              DM=34.5678901, DM_error=1.2e-4,
              flags={"be": "GUPPI", "nbin": 256, "snr": 123.4567, "phi_DM_cov": 1.23e-12,
                     "phs": 0.123456789, "flux": 1.234567, "skip": None})
    line = toa_string(toa)
    # format of pplib.py:3465-3497
    assert line == ("a.fits 0.00000000 55000.123456789012345   0.123  1 -pp_dm 34.5678901"
                    " -pp_dme 0.0001200 -be GUPPI -nbin 256 -snr 123.457"
                    " -phi_DM_cov 1.2e-12 -phs 0.12345679 -flux 1.23457")
    assert toa.snr == 123.4567 and toa.be == "GUPPI"


def test_dataportrait_from_arrays_field_names():
    from pulseportraiture_amd.pptoas import data_from_arrays
    rng = np.random.default_rng(0)
    sub = rng.normal(size=(2, 1, 6, 32))
    w = np.ones((2, 6)); w[0, 2] = 0
    d = data_from_arrays(sub, np.linspace(1100, 1900, 6), [0.003, 0.003], [55000.1, 55000.2],
                         weights=w, noise_stds=np.full((2, 1, 6), 0.1), DM=12.0)
    dp = pplib.DataPortrait(d)
    assert dp.port.shape == (6, 32) and dp.portx.shape == (5, 32)
    assert np.all(dp.port[2] == 0) and np.array_equal(dp.portx[2], sub[0, 0, 3])
    np.testing.assert_array_equal(dp.freqsxs[0], d.freqs[0, [0, 1, 3, 4, 5]])
    assert dp.noise_stdsxs.shape == (5,) and dp.SNRsxs.shape == (5,)
    assert dp.DM == 12.0 and dp.nbin == 32 and dp.nchan == 6 and list(dp.ok_isubs) == [0, 1]
    with np.testing.assert_raises(RuntimeError):
        pplib.DataPortrait("some.fits")


def test_spline_model_portraits_match_reference():
    from pulseportraiture_amd import splmodel
    g = np.load(os.path.join(GOLDEN, "spline_model_256.npz"))
    path = os.path.join(GOLDEN, "example.spl")
    name, src, dfile, mean_prof, eigvec, tck = splmodel.read_spline_model(path, quiet=True)
    assert name == "example_spline" and eigvec.shape == (256, 2)
    np.testing.assert_allclose(splmodel.read_spline_model(path, g["freqs"], None, True)[1],
                               g["port"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(splmodel.read_spline_model(path, g["freqs"], 512, True)[1],
                               g["port_512"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(
        splmodel.gen_spline_portrait(mean_prof, g["freqs"], eigvec[:, :0], tck),
        g["port_flat"], rtol=0, atol=0)


def test_instrumental_response_matches_reference():
    """Host-side template preparation of get_TOAs(add_instrumental_response=True):
    the per-channel response (rect + gauss + dispersive smearing) and the analytic
    Gaussian transform, against the reference's own arrays (pptoaslib.py:14-50,
    112-179) stored by make_golden_gettoas.py."""
    import os
    import numpy as np
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "gettoas_ird.npz"))
    # the functions are plain NumPy/SciPy; load the module without the HIP library
    src = open(os.path.join(here, "..", "pulseportraiture_amd", "pptoaslib.py")).read()
    ns = {"np": np}
    exec(src[src.index("def gaussian_profile_FT("):], ns)
    ok = np.where(g["weights"][0] > 0)[0]
    resp = ns["instrumental_response_port_FT"](
        g["subints"].shape[-1], g["freqs"][0][ok], float(g["out_ird_DM"]), float(g["Ps"][0]),
        [float(v) for v in g["out_ird_wids"]], [str(v) for v in g["out_ird_types"]])
    np.testing.assert_allclose(resp, g["out_ird_resp"], rtol=1e-13, atol=1e-15)
    gft = ns["gaussian_profile_FT"](g["subints"].shape[-1], 0.3, 0.02, 1.7)
    np.testing.assert_allclose(gft, g["out_ird_gauss_FT"], rtol=1e-13, atol=1e-13)


def test_tim_lines_match_the_hand_derived_fixture(tmp_path):
    """The .tim line format of write_TOAs (pplib.py:3445-3503), byte for byte against
    tests/golden/toa_lines.tim -- lines written out by hand from the reference's format
    strings (15-decimal MJD fraction spliced onto the integer day, three spaces, %.3f
    error, two spaces, the telescope code; -pp_dm / -pp_dme with 7 decimals; flags in
    insertion order with the per-type formats; None flags skipped; 0.0 for an infinite
    frequency).  Also through a file, append and overwrite."""
    import collections
    import json
    from pulseportraiture_amd.pptoas import MJD, TOA, toa_string, write_TOAs
    here = os.path.join(os.path.dirname(__file__), "golden")
    want = [ln.rstrip("\n") for ln in open(os.path.join(here, "toa_lines.tim")) if not ln.startswith("#")]
    recs = json.load(open(os.path.join(here, "toa_lines.json")))
    toas = []
    for r in recs:
        flags = collections.OrderedDict((k, v) for k, v in r["flags"])
        freq = np.inf if r["frequency"] == "inf" else r["frequency"]
        toas.append(TOA(r["archive"], freq, MJD(r["mjd_day"], r["mjd_frac"]), r["TOA_error"], r["telescope"],
                        r["telescope_code"], r["DM"], r["DM_error"], flags))
    assert [toa_string(t) for t in toas] == want
    # inf_is_zero=False writes "inf" like Python's %f would
    assert toa_string(toas[1], inf_is_zero=False).startswith("fake.fits inf 56001.000000000000001")
    out = tmp_path / "t.tim"
    write_TOAs(toas, outfile=str(out), append=False)
    write_TOAs(toas[0], outfile=str(out), append=True)
    assert open(out).read() == "\n".join(want + [want[0]]) + "\n"
    # the S/N cut drops the second TOA (snr 9.5) and any TOA without an snr flag
    write_TOAs(toas, SNR_cutoff=10.0, outfile=str(out), append=False)
    assert open(out).read() == want[0] + "\n"


def test_get_TOAs_flag_order_is_the_reference_insertion_order():
    """The flag dictionary get_TOAs builds follows the statement order of the reference
    (pptoas.py:607-657), which is the order the .tim line lists them in."""
    import inspect
    from pulseportraiture_amd import pptoas
    src = inspect.getsource(pptoas.GetTOAs.get_TOAs)
    order = ["'gm'", "'gm_err'", "'scat_time'", "'scat_ref_freq'", "'scat_ind'", "'scat_ind_err'", "'be'",
             "'fe'", "'f'", "'nbin'", "'nch'", "'nchx'", "'bw'", "'chbw'", "'subint'", "'tobs'", "'fratio'",
             "'tmplt'", "'snr'", "'phi_DM_cov'", "'gof'", "'phs'", "'phs_err'", "'flux'", "'flux_err'",
             "'flux_ref_freq'", "'par_angle'"]
    pos = [src.index("toa_flags[%s]" % k) for k in order]
    assert pos == sorted(pos), [k for k, a, b in zip(order[1:], pos, pos[1:]) if b < a]


def test_scale_run_checker(tmp_path):
    """tools/check_scale.py (the table tools/run_scale.sh writes on a multi-GPU node): accepts runs whose
    strong-scaling records are the same bits for every N, flags a run whose records differ or whose rows /
    return codes are off."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    total, nsub, steps = 40, 8, 2
    rng = np.random.default_rng(0)
    rec = rng.normal(size=(total, 18))
    rec[:, 17] = 2.0

    def write(n, mode, records=None, rows=None, rc=2):
        line = {"metric": "subint_fits_per_sec", "value": 1000.0 * n, "ms_per_step": 1.0, "n_gpus": n, "scaling": mode,
                "steps": steps}
        if mode == "weak":
            line["config"] = {"nsub_per_gpu_per_step": nsub}
            line["gathered_records"] = {"rows": rows if rows is not None else n * nsub * steps}
            line["convergence"] = {"return_codes": {str(rc): n * nsub}}
        else:
            base, extra = divmod(total, n)
            line["config"] = {"total_nsub": total, "fits_per_rank": [base + (1 if r < extra else 0) for r in range(n)]}
            line["gathered_records"] = {"rows": total, "return_code_sum": float(records[:, 17].sum())}
            line["max_abs_dDM_over_err"] = 3.0
            np.save(tmp_path / ("records_strong_n%d.npy" % n), records)
        (tmp_path / ("%s_n%d.json" % (mode, n))).write_text("some banner\n" + json.dumps(line) + "\n")

    for n in (1, 2):
        write(n, "weak")
        write(n, "strong", rec)
    cmd = [sys.executable, os.path.join(root, "tools", "check_scale.py"), str(tmp_path), "1", "2"]
    ok = subprocess.run(cmd, capture_output=True, text=True)
    assert ok.returncode == 0, ok.stdout + ok.stderr
    assert "bit for bit" in ok.stdout and ok.stdout.count("OK") == 4
    moved = rec.copy()
    moved[7, 0] += 1e-12
    write(2, "strong", moved)
    bad = subprocess.run(cmd, capture_output=True, text=True)
    assert bad.returncode == 1 and "differ from the N = 1 job in 1 rows" in bad.stdout
    write(2, "strong", rec)
    write(2, "weak", rows=5)
    bad = subprocess.run(cmd, capture_output=True, text=True)
    assert bad.returncode == 1 and "FAIL" in bad.stdout
