#!/bin/bash
make -B -C pulseportraiture_amd/csrc EXTRA=-DPP_XSPEC_STAMPS=1 >/dev/null 2>&1 || { echo build failed; exit 1; }
python tools/dev_xspec_stamps.py "$@"
make -B -C pulseportraiture_amd/csrc >/dev/null 2>&1
