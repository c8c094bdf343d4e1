import sys, os, json
import numpy as np
sys.path.insert(0, "/root/repo")
from tests import test_gpu_parity as T
from pulseportraiture_amd.pptoas import GetTOAs, MJD, data_from_arrays
name = "gettoas_opt_scatflux"
g = T._load(name)
q = lambda k: g["in0_%s" % k]
epochs = [MJD(int(d), float(f)) for d, f in zip(q("epoch_days"), q("epoch_fracs"))]
b = data_from_arrays(q("subints"), q("freqs"), q("Ps"), epochs, weights=q("weights"), noise_stds=q("noise_stds"),
    SNRs=q("SNRs"), DM=float(q("scal_DM")), doppler_factors=q("doppler_factors"), backend_delay=float(q("scal_backend_delay")),
    telescope=str(q("scal_telescope")), telescope_code=str(q("scal_telescope_code")), backend=str(q("scal_backend")),
    frontend=str(q("scal_frontend")), bw=float(q("scal_bw")), nu0=float(q("scal_nu0")), subtimes=q("subtimes"),
    source=str(q("scal_source")), filename=str(q("filename")))
gt = GetTOAs(b, os.path.join(T.GOLDEN, "example.gmodel"), quiet=True)
gt.get_TOAs(quiet=True, seed='reference', fit_scat=True, print_flux=True, scat_guess=(30e-6, 1500.0, -4.0))
ok = g["out_a0_ok_isubs"].astype(int)
for f in ("phis", "DMs", "taus", "alphas", "nu_refs", "fluxes", "flux_errs", "flux_freqs", "red_chi2s", "nfevals"):
    a = np.asarray(getattr(gt, f)[0], dtype=float)[ok]; w = g["out_a0_" + f][ok]
    print(f, np.abs(a - w).max() if a.ndim == 1 else np.abs(a - w).max(axis=0), "rel", (np.abs(a - w) / np.maximum(np.abs(w), 1e-300)).max())
print(np.asarray(gt.nfevals[0]), g["out_a0_nfevals"])
