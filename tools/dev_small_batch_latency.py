"""Latency of SMALL batches on a wide band: channel chunks of the evaluators are a function of the band
alone (PP_CHUNK_CHANNELS = 256, whatever the batch: a subint's answer must not depend on its neighbours), so a lone
4096-channel subint launches 16 evaluator workgroups where round 4 launched up to 64.  Times 1 / 2 / 4 / 8 subints of
4096 x 2048 and 2048 x 2048, scattering (evaluation loop) and phase + DM (one pass), device-resident portraits, best of
5 calls:   PP_TOAS_LIB=variants/chunk64.so python tools/dev_small_batch_latency.py    for the 64-channel chunk build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _full_shape_case

print("library:", os.environ.get("PP_TOAS_LIB", "product"))
for C, flags, l10, tau in ((4096, [1, 1, 0, 1, 1], True, 20.0), (2048, [1, 1, 0, 1, 1], True, 20.0), (4096, [1, 1, 0, 0, 0], False, None)):
    for nsub in (1, 2, 4, 8):
        e, data, freqs, model, P, x0, errs, nu_fit, kw = _full_shape_case(C, 2048, flags, l10, nsub=nsub, tau_us=tau, seed=3)
        best = 1e9
        for rep in range(6):
            t0 = time.perf_counter()
            r = e.fit_batch(data, freqs, P, x0, **kw)
            best = min(best, time.perf_counter() - t0) if rep else best
        print("C %4d flags %s nsub %d: %.3f ms per call (device %.3f ms), nfeval %s, checksum %.15g" %
              (C, "".join(map(str, flags)), nsub, 1e3 * best, 1e3 * r["duration"], r["nfeval"].tolist()[:2], float(np.sum(r["params"]))))
        e.close()
