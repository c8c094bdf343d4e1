#!/usr/bin/env python3
"""Goldens for the fit_flags families of get_nu_zeros the main set does not reach
(pptoaslib.py:753-767, 837-892): [1,0,1,0,0] (phase + GM), [0,0,0,1,1] (tau +
alpha only) and [1,1,1,1,0] (no alpha; quintic / quartic zero-covariance
polynomial, option 0 and 1).  Same machinery as make_golden.py: the true
reference imported in a scratch directory, build container only."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    ref, tmp = mg.import_reference()
    S = mg.SEED
    mg.fit_case(ref, "fpf_64x256_phiGM", 64, 256, S + 13, [1, 0, 1, 0, 0], GM=0.25)
    mg.fit_case(ref, "fpf_64x256_taualpha", 64, 256, S + 14, [0, 0, 0, 1, 1], log10_tau=True,
                tau_us=20.0)
    mg.fit_case(ref, "fpf_64x256_taualpha_lin", 64, 256, S + 15, [0, 0, 0, 1, 1],
                log10_tau=False, tau_us=30.0)
    mg.fit_case(ref, "fpf_64x256_phiDMGMtau", 64, 256, S + 16, [1, 1, 1, 1, 0], log10_tau=True,
                tau_us=20.0, GM=0.25)
    mg.fit_case(ref, "fpf_64x256_phiDMGMtau_opt1", 64, 256, S + 16, [1, 1, 1, 1, 0],
                log10_tau=True, tau_us=20.0, GM=0.25, option=1)
    mg.fit_case(ref, "fpf_64x256_phiDMGMtau_lin", 64, 256, S + 17, [1, 1, 1, 1, 0],
                log10_tau=False, tau_us=30.0, GM=0.25)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
