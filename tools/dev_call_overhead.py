import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import numpy as np, torch
from pulseportraiture_amd.engine import Engine
from pulseportraiture_amd import gmodel
from pulseportraiture_amd.pplib import guess_fit_freq
C, B, nsub = 512, 1024, 1024
eng = Engine(0)
freqs, model, P0 = gmodel.example_model(C, B)
eng.set_model(model)
data = torch.empty((nsub, C, B), dtype=torch.float64, device="cuda:0")
inj = np.zeros((nsub, 3)); inj[:, 1] = 34.5
P = np.full(nsub, P0)
eng.synth_portraits(data, freqs, P, inj, 0.05, 1, 0)
x0 = np.zeros((nsub, 5)); x0[:, 1] = 34.5
nu_fit = float(guess_fit_freq(freqs))
errs = torch.full((nsub, C), 0.05, dtype=torch.float64, device="cuda:0")
recs = torch.zeros((nsub, 18), dtype=torch.float64, device="cuda:0")
from pulseportraiture_amd.pplib import Dconst
x0[:, 0] = (Dconst * 34.5 / P0 / nu_fit ** 2) % 1.0
kw = dict(errs=errs, nu_fits=np.full((nsub, 3), nu_fit), fit_flags=[1, 1, 0, 0, 0], per_channel="device", records=recs)
for _ in range(5): r = eng.fit_batch(data, freqs, P, x0, **kw)
print("return codes", np.unique(r["return_code"]), "npass", r["npass"].max(), "device ms", 1e3 * r["duration"])
for prof, flush in ((0, 0), (0, 1), (1, 0), (1, 1)):
    eng.set_option("profile", prof)
    eng.set_option("eager_flush", flush)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): r = eng.fit_batch(data, freqs, P, x0, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("eager_flush=%d" % flush, end=" ")
    print("profile=%d: %.3f ms per call, device %.3f ms -> host overhead %.3f ms" % (prof, 1e3 * dt, 1e3 * r["duration"], 1e3 * (dt - r["duration"])))
eng.set_option("profile", 0)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): r = eng.fit_batch(data, freqs, P, x0, **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])

# enqueue / collect two deep
eng.set_option("profile", 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(200):
    eng.enqueue(data, freqs, P, x0, **kw)
    if k: r = eng.collect()
r = eng.collect()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print("enqueue/collect two deep: %.3f ms per batch" % (1e3 * dt))
