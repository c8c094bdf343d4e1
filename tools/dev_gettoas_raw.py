"""Caller-level raw parity: GetTOAs.get_TOAs(seed='reference') against the reference's own
get_TOAs outputs (tests/golden/gettoas_*.npz).  (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _load, GOLDEN, _dphi_arr
from pulseportraiture_amd.pptoas import GetTOAs, MJD, data_from_arrays

for name in ["gettoas_phiDM", "gettoas_phiDM_nurefs", "gettoas_GM", "gettoas_scat", "gettoas_zap", "gettoas_ird"]:
    g = _load(name)
    epochs = [MJD(int(d), float(f)) for d, f in zip(g["epoch_days"], g["epoch_fracs"])]
    data = data_from_arrays(
        g["subints"], g["freqs"], g["Ps"], epochs, weights=g["weights"],
        noise_stds=g["noise_stds"], SNRs=g["SNRs"], DM=float(g["scal_DM"]),
        doppler_factors=g["doppler_factors"],
        backend_delay=float(g["scal_backend_delay"]), telescope=str(g["scal_telescope"]),
        telescope_code=str(g["scal_telescope_code"]), backend=str(g["scal_backend"]),
        frontend=str(g["scal_frontend"]), bw=float(g["scal_bw"]), nu0=float(g["scal_nu0"]),
        subtimes=g["subtimes"], source=str(g["scal_source"]), filename="fake.fits")
    kw = {}
    for k in g.files:
        if k.startswith("kw_"):
            v = g[k]
            kw[k[3:]] = v.item() if v.ndim == 0 else tuple(v.tolist())
    gt = GetTOAs(data, os.path.join(GOLDEN, "example.gmodel"), quiet=True)
    if "out_ird_DM" in g.files:
        gt.instrumental_response_dict = gt.ird = {
            'DM': float(g["out_ird_DM"]), 'wids': [float(v) for v in g["out_ird_wids"]],
            'irf_types': [str(v) for v in g["out_ird_types"]]}
    gt.get_TOAs(quiet=True, seed='reference', **kw)
    ok = g["out_ok_isubs"]
    print(name, kw)
    print("   dphi  max %.2e" % _dphi_arr(np.asarray(gt.phis[0])[ok], g["out_phis"][ok]).max(),
          " dDM max %.2e" % np.abs(np.asarray(gt.DMs[0])[ok] - g["out_DMs"][ok]).max(),
          " dGM max %.2e" % np.abs(np.asarray(gt.GMs[0])[ok] - g["out_GMs"][ok]).max(),
          " dtau max %.2e" % np.abs(np.asarray(gt.taus[0])[ok] - g["out_taus"][ok]).max(),
          " dalpha max %.2e" % np.abs(np.asarray(gt.alphas[0])[ok] - g["out_alphas"][ok]).max())
    print("   nfev", np.asarray(gt.nfevals[0])[ok], " ref", g["out_nfevals"][ok] if "out_nfevals" in g.files else None)
    print("   nu_refs rel max %.2e" % np.nanmax(np.abs(np.array(gt.nu_refs[0])[ok] / g["out_nu_refs"][ok] - 1)))
