#!/usr/bin/env python3
"""Goldens at row lengths that are NO power of two (numpy.fft.rfft takes every nbin,
pptoaslib.py:976-979): fit_portrait_full of the TRUE reference at nbin = 1000, 100 and 1536,
inputs, objective points and all 25 outputs, exactly as tests/golden/make_golden.py writes the
power-of-two cases (same helpers; build container only).

    python tests/golden/make_golden_nbin.py        # writes tests/golden/fpf_*x{1000,100,1536}_*.npz
"""
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    ref, tmp = mg.import_reference()
    S = mg.SEED + 500
    mg.fit_case(ref, "fpf_48x1000_phiDM", 48, 1000, S + 1, [1, 1, 0, 0, 0], DM0=34.56789)
    mg.fit_case(ref, "fpf_48x1000_scat", 48, 1000, S + 2, [1, 1, 0, 1, 1], log10_tau=True, tau_us=20.0)
    mg.fit_case(ref, "fpf_40x100_phiDMGM", 40, 100, S + 3, [1, 1, 1, 0, 0], GM=0.25)
    mg.fit_case(ref, "fpf_24x1536_phiDMtau", 24, 1536, S + 4, [1, 1, 0, 1, 0], log10_tau=False, tau_us=30.0)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
