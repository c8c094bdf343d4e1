import sys, os, json
import numpy as np
sys.path.insert(0, "/root/repo")
from tests import test_gpu_parity as T
names = ["gettoas_phiDM", "gettoas_phiDM_nurefs", "gettoas_GM", "gettoas_scat", "gettoas_zap", "gettoas_ird"]
for name in names:
    g = T._load(name)
    gt, kw = T._gettoas_from_golden(g)
    gt.get_TOAs(quiet=True, seed='reference', **kw)
    nf = np.asarray(gt.nfevals[0]); ref = g["out_nfevals"]
    print(name, "device", nf.tolist(), "reference", ref.tolist(), "diff at", np.where(nf != ref)[0].tolist())
from pulseportraiture_amd.pptoas import GetTOAs, MJD, data_from_arrays
for name in T.OPTION_GOLDENS:
    g = T._load(name)
    narch = int(g["narchives"])
    bunches = []
    for ia in range(narch):
        q = lambda k: g["in%d_%s" % (ia, k)]
        epochs = [MJD(int(d), float(f)) for d, f in zip(q("epoch_days"), q("epoch_fracs"))]
        bunches.append(data_from_arrays(
            q("subints"), q("freqs"), q("Ps"), epochs, weights=q("weights"), noise_stds=q("noise_stds"),
            SNRs=q("SNRs"), DM=float(q("scal_DM")), doppler_factors=q("doppler_factors"),
            backend_delay=float(q("scal_backend_delay")), telescope=str(q("scal_telescope")),
            telescope_code=str(q("scal_telescope_code")), backend=str(q("scal_backend")),
            frontend=str(q("scal_frontend")), bw=float(q("scal_bw")), nu0=float(q("scal_nu0")),
            subtimes=q("subtimes"), source=str(q("scal_source")), filename=str(q("filename"))))
    kw = {}
    for k in g.files:
        if k.startswith("kw_"):
            v = g[k]
            kw[k[3:]] = v.item() if v.ndim == 0 else tuple(v.tolist())
    gt = GetTOAs(bunches if narch > 1 else bunches[0], os.path.join(T.GOLDEN, "example.gmodel"), quiet=True)
    gt.get_TOAs(quiet=True, seed='reference', **kw)
    for ia in range(narch):
        nf = np.asarray(gt.nfevals[ia]); ref = g["out_a%d_nfevals" % ia].astype(int)
        print(name, ia, "device", nf.tolist(), "reference", ref.tolist(), "diff at", np.where(nf != ref)[0].tolist())
