"""Per-workgroup wall time of k_xspec on the headline shape (GPU box; needs a
library built with EXTRA=-DPP_XSPEC_STAMPS=1):
  make -B -C pulseportraiture_amd/csrc EXTRA=-DPP_XSPEC_STAMPS=1
  python tools/dev_xspec_stamps.py [nsub] [nchan] [nbin]
Prints when the workgroups of the last launch started and ended (100 MHz wall
clock), the shader clock each one saw, and the spread per XCD."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, ".")
from pulseportraiture_amd import gmodel, _lib
from pulseportraiture_amd.engine import default_engine
import torch

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
GOLDEN = os.path.join("tests", "golden")
mdl = gmodel.read_gmodel(os.path.join(GOLDEN, "example.gmodel"))
freqs = np.linspace(1200.0, 1800.0, C, endpoint=False) + 300.0 / C
P = 0.005
eng = default_engine()
eng.set_model_gaussian(mdl, freqs, B, P, slot=0)
data = torch.empty((nsub, C, B), dtype=torch.float64, device="cuda:0")
rng = np.random.default_rng(1)
inj = np.zeros((nsub, 3)); inj[:, 0] = rng.uniform(-0.4, 0.4, nsub); inj[:, 1] = 30.0 + rng.normal(0, 1e-4, nsub)
eng.synth_portraits(data, freqs, np.full(nsub, P), inj, 0.05, 7, 0)
x0 = np.zeros((nsub, 5)); x0[:, 0] = inj[:, 0]; x0[:, 1] = 30.0
errs = torch.full((nsub, C), 0.05, dtype=torch.float64, device="cuda:0")
lib = ctypes.CDLL(_lib.LIB_PATH)
for rep in range(6):
    res = eng.fit_batch(data, freqs, np.full(nsub, P), x0, errs=errs, per_channel="device")
print("fit duration %.2f ms" % (1e3 * res["duration"]))
NWG = 2048
buf = (ctypes.c_ulonglong * (6 * NWG))()
rc = lib.pp_debug_xspec_stamps(buf, NWG)
assert rc == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(NWG, 6).astype(np.int64)
t0 = st[:, 0].min()
beg = (st[:, 0] - t0) / 100.0       # us
end = (st[:, 1] - t0) / 100.0
dur = end - beg
clk = (st[:, 3] - st[:, 2]) / (dur * 1e-6) / 1e9
xcc = st[:, 4] & 15
print("launch spread of starts: %.1f us; ends: min %.0f median %.0f max %.0f us" % (beg.max(), end.min(), np.median(end), end.max()))
print("duration: min %.0f mean %.0f max %.0f us -> mean/max %.3f" % (dur.min(), dur.mean(), dur.max(), dur.mean() / end.max()))
print("shader clock GHz: min %.3f mean %.3f max %.3f" % (clk.min(), clk.mean(), clk.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCD %d: %4d workgroups, end %.0f..%.0f us (mean %.0f), clock %.3f GHz" % (
            x, m.sum(), end[m].min(), end[m].max(), end[m].mean(), clk[m].mean()))
q = np.percentile(end, [1, 10, 25, 50, 75, 90, 99])
print("end percentiles 1/10/25/50/75/90/99 %%: %s" % " ".join("%.0f" % v for v in q))
