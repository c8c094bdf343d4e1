"""Where align_subints and the oracle loop part: seeds, fits, accumulation (run on the GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pptoas_oracle as orc
from pulseportraiture_amd.engine import Engine
from pulseportraiture_amd.pplib import Dconst
from tests.synth_host import model_portrait

eng = Engine(0)
C_, nbin, nsub, sigma, fit_dm = 16, 256, 6, 0.05, True
freqs, model = model_portrait(C_, nbin)
rng = np.random.default_rng(77)
Ps = np.full(nsub, 0.0031) * (1 + 1e-6 * np.arange(nsub))
ports = np.zeros((nsub, C_, nbin))
for i in range(nsub):
    rot = orc.rotate_portrait_full(model * rng.uniform(0.7, 1.5), -rng.uniform(-0.5, 0.5),
                                   -(rng.normal(0, 3e-4) if fit_dm else 0.0), 0.0, freqs, np.inf, np.inf, Ps[i])
    ports[i] = rot + rng.normal(0, sigma, rot.shape)
weights = np.ones((nsub, C_)); weights[1, [2, 9]] = 0.0; weights[4, :3] = 0.0
weights[5, :] = 0.0; weights[5, 7] = 1.0
errs = np.full((nsub, C_), sigma)
snrs = rng.uniform(5, 50, (nsub, C_))
tmpl = orc.rotate_data(model, 0.013) * 0.8
mask = (weights > 0).astype(np.uint8)
f2 = np.broadcast_to(freqs, (nsub, C_))
for it in range(2):
    print('iteration', it + 1)
    eng.set_model(tmpl)
    nu_fit = np.array([orc.guess_fit_freq(freqs[mask[i] > 0], snrs[i][mask[i] > 0]) for i in range(nsub)])
    mprofs = np.array([tmpl[mask[i] > 0].mean(axis=0) for i in range(nsub)])
    seed = eng.reference_phase_seed(ports, f2, Ps, weights, mprofs, phi=np.zeros(nsub), DM=np.zeros(nsub), nu_DM=np.inf,
                                    Ns=nbin, finish='simplex')
    x0 = np.zeros((nsub, 5)); x0[:, 0] = seed[:, 0]
    res = eng.fit_batch(ports, f2, Ps, x0, errs=errs, chan_mask=mask, nu_fits=np.repeat(nu_fit[:, None], 3, axis=1),
                        fit_flags=[1, 1, 0, 0, 0], log10_tau=False, method='trust-ncg')
    acc_o = np.zeros((C_, nbin)); tw_o = np.zeros(C_)
    for i in range(nsub):
        ich = np.where(weights[i] > 0)[0]
        if len(ich) < 2:
            r = orc.fit_phase_shift(ports[i, ich[0]], tmpl[ich[0]], errs[i, ich[0]], Ns=nbin)
            r1 = eng.fit_phase_shift_batch(ports[i, ich], tmpl[ich], noise=errs[i, ich], Ns=nbin, finish='simplex')
            print("single", i, r.phase, r1[0, 0], r.phase - r1[0, 0], r.scale, r1[0, 2])
            continue
        rp = orc.rotate_data(ports[i, ich], 0.0, 0.0, Ps[i], freqs[ich], nu_fit[i])
        g = orc.fit_phase_shift(np.average(rp, axis=0, weights=weights[i, ich]), tmpl[ich].mean(axis=0), Ns=nbin).phase
        r = orc.fit_portrait_full(ports[i, ich], tmpl[ich], [g, 0.0, 0.0, 0.0, 0.0], Ps[i], freqs[ich], [nu_fit[i]] * 3,
                                  [None] * 3, errs[i, ich], [1, 1, 0, 0, 0], log10_tau=False)
        print(i, "seed d=%.2e" % (g - seed[i, 0]), "phi d=%.2e DM d=%.2e nu d=%.2e scale d=%.2e nfev %d/%d" % (
            r.phi - res["params"][i, 0], r.DM - res["params"][i, 1], r.nu_DM - res["nu_refs"][i, 0],
            np.abs(r.scales - res["scales"][i, ich]).max(), r.nfeval, res["nfeval"][i]))
        w = r.scales / errs[i, ich] ** 2
        rot_o = orc.rotate_data(ports[i, ich], r.phi, r.DM, Ps[i], freqs[ich], r.nu_DM)
        wz = np.zeros((1, C_)); wz[0, ich] = w
        a1, t1 = eng.align_accumulate(ports[i:i + 1], f2[i:i + 1], Ps[i:i + 1], np.array([r.phi]), np.array([r.DM]),
                                      np.array([r.nu_DM]), wz)
        acc_o[ich] += w[:, None] * rot_o
        tw_o[ich] += w
        print("   accumulate d=%.2e (peak %.2f)" % (np.abs(a1[ich] - w[:, None] * rot_o).max(), np.abs(w[:, None] * rot_o).max()))
    tmpl = acc_o / tw_o[:, None]
