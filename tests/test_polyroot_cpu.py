"""The device's polynomial root selection (pick_poly_root in csrc/pp_kernels.h: the
np.roots + selection of get_nu_zeros, pptoaslib.py:791-794, 859-863) compiled for the
host with g++ and compared with numpy.roots -- on the coefficient sets the reference
forms for the [1,1,1,0,0] and [1,1,1,1,0] goldens (ratios of 1e17 between
coefficients, one root at zero to rounding) and on random polynomials of degree 3-6
with roots from 1e-3 to 1e7."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "pulseportraiture_amd", "csrc", "pp_kernels.h")


@pytest.fixture(scope="module")
def polyroot(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = open(HDR).read()
    i = src.index("__device__ inline double pick_poly_root")
    j = src.index("// post-fit stage")
    j = src.rindex("// ----", 0, j)
    tmp = tmp_path_factory.mktemp("polyroot")
    cpp = tmp / "poly.cpp"
    cpp.write_text("#include <cmath>\n#include <cstdio>\n#include <cstdlib>\n#define __device__\n"
                   "#define PP_TWO_PI 6.283185307179586476925286766559\nusing std::fmax; using std::fabs;\n" +
                   src[i:j] +
                   "\nint main(){ int deg, sq; double c[8], t;\n"
                   " while (scanf(\"%d %d %lf\", &deg, &sq, &t) == 3) { for (int i = 0; i <= deg; ++i) if (scanf(\"%lf\", &c[i]) != 1) return 1;\n"
                   "  printf(\"%.17g\\n\", pick_poly_root(c, deg, t, sq)); } return 0; }\n")
    exe = tmp / "poly"
    subprocess.run(["g++", "-O2", "-o", str(exe), str(cpp)], check=True)

    def run(cases):
        text = "".join("%d %d %.17g %s\n" % (len(c) - 1, int(sq), t, " ".join("%.17g" % v for v in c))
                       for c, t, sq in cases)
        out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout
        return np.array([float(v) for v in out.split()])
    return run


def test_reference_coefficient_sets(polyroot):
    from oracle import pptoas_oracle as orc
    golden = os.path.join(ROOT, "tests", "golden")
    calls, orig = [], orc._pick_root

    def spy(coeffs, target, sqrt=False):
        calls.append(([float(v) for v in coeffs], float(target), bool(sqrt)))
        return orig(coeffs, target, sqrt)
    orc._pick_root = spy
    try:
        for n in ("phiDMGMtau", "phiDMGMtau_lin", "phiDMGMtau_opt1", "phiDMGM", "phiDMGM_opt1"):
            g = np.load(os.path.join(golden, "fpf_64x256_%s.npz" % n))
            kw = dict(option=int(g["option"]), is_toa=bool(g["is_toa"])) if "option" in g.files else {}
            orc.fit_portrait_full(g["data"], g["model"], g["init_params"], float(g["P"]), g["freqs"],
                                  list(g["nu_fits"]), [None] * 3, g["errs"], list(g["fit_flags"]),
                                  log10_tau=bool(g["log10_tau"]), **kw)
    finally:
        orc._pick_root = orig
    want = np.array([orig(c, t, s) for c, t, s in calls])
    np.testing.assert_allclose(polyroot(calls), want, rtol=1e-13)


def test_random_polynomials(polyroot):
    rng = np.random.default_rng(0)
    cases, want = [], []
    while len(cases) < 400:
        deg = int(rng.integers(3, 7))
        scale = 10 ** rng.uniform(-3, 7)
        nreal = int(rng.integers(1, deg + 1))
        if (deg - nreal) % 2:
            nreal += 1
        roots = list(scale * rng.uniform(0.1, 3, nreal) * rng.choice([1, 1, 1, -1], nreal))
        for _ in range((deg - nreal) // 2):
            z = scale * (rng.normal() + 1j * rng.uniform(0.3, 2))
            roots += [z, np.conj(z)]
        pos = [r.real for r in roots if abs(r.imag) == 0 and r.real > 0]
        if not pos:
            continue
        target = scale * rng.uniform(0.1, 3)
        cases.append((list(np.real(np.poly(roots)) * 10 ** rng.uniform(-20, 20)), target, False))
        want.append(min(pos, key=lambda v: abs(target - v)))
    np.testing.assert_allclose(polyroot(cases), want, rtol=1e-9)
