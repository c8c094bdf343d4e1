"""Are the bench's phase guesses (rotate, channel mean, fit_phase_shift with the simplex
finish) the same from call to call and from process to process?  (GPU box)"""
import sys, hashlib
import numpy as np
import torch
sys.path.insert(0, ".")
from pulseportraiture_amd.engine import Engine
from pulseportraiture_amd import gmodel

eng = Engine(0)
C, B, nsub = 4096, 2048, 1024
freqs, model, P0 = gmodel.example_model(C, B)
eng.set_model(model)
P = np.full(nsub, P0)
rng = np.random.default_rng(20260101)
inj = np.zeros((nsub, 3)); inj[:, 0] = rng.uniform(-0.5, 0.5, nsub); inj[:, 1] = 34.56789 + rng.normal(3e-4, 2e-4, nsub)
data = torch.empty((nsub, C, B), dtype=torch.float32, device="cuda:0")
eng.synth_portraits(data, freqs, P, inj, 0.05, 20260101, 0)
seed_prof = model.mean(axis=0)
nu_mean = float(freqs.mean())
for rep in range(3):
    profs = np.empty((nsub, B))
    for s0 in range(0, nsub, 128):
        chunk = data[s0:s0 + 128].to(torch.float64).clone()
        eng.rotate_portraits(chunk, freqs, P[s0:s0 + 128], DM=np.full(chunk.shape[0], 34.56789), nu_DM=nu_mean)
        profs[s0:s0 + 128] = chunk.mean(dim=1).cpu().numpy()
    outs = {}
    for fin in ("simplex", "newton"):
        outs[fin] = eng.fit_phase_shift_batch(profs, seed_prof, Ns=100, finish=fin)[:, 0]
    d = np.abs(outs["simplex"] - outs["newton"])
    print("rep", rep, "profs md5", hashlib.md5(profs.tobytes()).hexdigest()[:10],
          "simplex md5", hashlib.md5(outs["simplex"].tobytes()).hexdigest()[:10],
          "newton md5", hashlib.md5(outs["newton"].tobytes()).hexdigest()[:10],
          "| simplex-newton: max %.2e at %d, #>1e-4: %d" % (d.max(), d.argmax(), (d > 1e-4).sum()),
          " [207] %.3e [694] %.3e" % (d[207], d[694]))
