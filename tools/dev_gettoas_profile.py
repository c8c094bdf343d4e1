"""Where does GetTOAs.get_TOAs spend its time on a many-subint archive?  (GPU box)
  python tools/dev_gettoas_profile.py [nsub] [nchan] [nbin]"""
import cProfile, pstats, sys, time, os
import numpy as np
sys.path.insert(0, ".")
from pulseportraiture_amd import gmodel
from pulseportraiture_amd.engine import default_engine
from pulseportraiture_amd.pptoas import GetTOAs, MJD, data_from_arrays
import torch

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
GOLDEN = os.path.join("tests", "golden")
mdl = gmodel.read_gmodel(os.path.join(GOLDEN, "example.gmodel"))
freqs = np.linspace(1200.0, 1800.0, C, endpoint=False) + 300.0 / C
P = 0.005
eng = default_engine()
eng.set_model_gaussian(mdl, freqs, B, P, slot=0)
data = torch.empty((nsub, C, B), dtype=torch.float64, device="cuda:0")
rng = np.random.default_rng(1)
inj = np.zeros((nsub, 3)); inj[:, 0] = rng.uniform(-0.4, 0.4, nsub); inj[:, 1] = 30.0 + rng.normal(0, 1e-4, nsub)
eng.synth_portraits(data, freqs, np.full(nsub, P), inj, 0.02, 7, 0)
sub = data.cpu().numpy()[:, None]
epochs = [MJD(58000 + i, 0.25) for i in range(nsub)]
d = data_from_arrays(sub, np.tile(freqs, (nsub, 1)), np.full(nsub, P), epochs, weights=np.ones((nsub, C)),
                     noise_stds=np.full((nsub, 1, C), 0.02), SNRs=np.full((nsub, 1, C), 10.0), DM=30.0,
                     doppler_factors=np.ones(nsub), backend_delay=0.0, telescope="GBT", telescope_code="1",
                     backend="X", frontend="Y", bw=600.0, nu0=1500.0, subtimes=np.full(nsub, 10.0),
                     source="fake", filename="fake.fits")
gt = GetTOAs(d, os.path.join(GOLDEN, "example.gmodel"), quiet=True)
gt.get_TOAs(quiet=True)                 # warm-up
gt = GetTOAs(d, os.path.join(GOLDEN, "example.gmodel"), quiet=True)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
gt.get_TOAs(quiet=True)
pr.disable()
dt = time.perf_counter() - t0
print("get_TOAs: %.1f ms for %d subints of %d x %d (%.0f subints/s); device fit %.1f ms" % (
    dt * 1e3, nsub, C, B, nsub / dt, 1e3 * np.sum(gt.fit_durations[0]) if hasattr(gt, "fit_durations") else -1))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
