#!/usr/bin/env python3
"""Goldens for the two switches of fit_portrait_full the main set leaves at their
defaults: option=1 (the other root of the zero-covariance polynomial for
phi+DM+GM fits, pptoaslib.py:779-812) and is_toa=False (nu_GM is not forced onto
nu_DM, pptoaslib.py:1048-1050).  Same machinery as make_golden.py (true reference
imported in a scratch directory; build container only)."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    ref, tmp = mg.import_reference()
    S = mg.SEED
    mg.fit_case(ref, "fpf_64x256_phiDMGM_opt1", 64, 256, S + 3, [1, 1, 1, 0, 0], GM=0.25, option=1)
    mg.fit_case(ref, "fpf_64x256_phiDMGM_notoa", 64, 256, S + 3, [1, 1, 1, 0, 0], GM=0.25,
                is_toa=False)
    mg.fit_case(ref, "fpf_64x256_all5_notoa", 64, 256, S + 10, [1, 1, 1, 1, 1], log10_tau=True,
                tau_us=20.0, GM=0.25, is_toa=False)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
