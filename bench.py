#!/usr/bin/env python3
"""Benchmark of the wideband-TOA hot path: subint fits/s on synthetic portraits
resident in HBM (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N ...          (starts the N ranks itself: self_launch below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the whole fit (rFFT, cross-spectrum, solve, zero-covariance
frequency, errors, S/N, chi2) over one batch of `nsub` subints per GPU; the batch is
generated on the device and its phase guesses are formed the way pptoas forms them
(dedispersed mean profile -> fit_phase_shift -> phase_transform, pptoas.py:421-457)
before the timed region.  The TOA records of every step stay in HBM and are gathered
ONCE, at the end of the timed region (RCCL gather of device tensors).  Rank 0 prints
ONE JSON line; at N = 1 it also carries the other workloads (--other-steps timed
steps after --other-warmup untimed ones each, same process), the CPU baseline of the
headline shape and of configs[1..3] (1 core and a one-worker-per-core pool each), and,
as its LAST key, a compact {workload: fits/s} summary of everything it measured.

    --total-nsub 100000   configs[4] as written: the subints are dealt in contiguous
                          shards to the ranks (strong scaling) and fitted in
                          device-generated sub-batches of --nsub
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nchan, nbin, fit_flags, log10_tau, default nsub per GPU, note)
    "toa-4096x2048-phiDM": (4096, 2048, [1, 1, 0, 0, 0], False, 1024,
                            "BASELINE.json target shape (configs[4] per-GPU "
                            "shard): 4096 chan x 2048 bin, phase+DM"),
    "cfg2-512x1024-phiDM": (512, 1024, [1, 1, 0, 0, 0], False, 1024,
                            "configs[1]: 1024 subints, 512 chan x 1024 bin, "
                            "phase+DM"),
    "cfg3-4096x2048-phiDMGM": (4096, 2048, [1, 1, 1, 0, 0], False, 1024,
                               "configs[2] shape: 4096 chan x 2048 bin, "
                               "phase+DM+GM"),
    "cfg4-2048x2048-scat": (2048, 2048, [1, 1, 0, 1, 1], True, 512,
                            "configs[3] shape: 2048 chan x 2048 bin, "
                            "phase+DM+tau+alpha"),
    "cfg1-64x256-phiDM": (64, 256, [1, 1, 0, 0, 0], False, 1,
                          "configs[0]: single 64 x 256 subint"),
}
# SURVEY 8(d)'s other regimes of a workload (Batch.__init__'s `variant`)
VARIANTS = {"lowsnr_sigma1.5": dict(sigma=1.5), "scint": dict(scint=True), "measured_noise": dict(measured_noise=True),
            "masked20": dict(mask_frac=0.2)}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s copy)
DCONST = 0.000241 ** -1


def injected_params(seed, first, n, flags, dm0, dm_offset, block=1024):
    """Injected (phi, DM, GM) of subints [first, first + n) of a job, and a unit-normal
    draw per subint for experiments: a function of the GLOBAL subint index only (blocks
    of `block` indices share one generator keyed on (seed, block number)), so shards,
    sub-batches and ranks can cut the job anywhere and still make the same subints."""
    inj = np.zeros((n, 3))
    extra = np.zeros(n)
    b0, b1 = first // block, (first + n - 1) // block
    for b in range(b0, b1 + 1):
        rng = np.random.default_rng([seed, b])
        phi = rng.uniform(-0.5, 0.5, block)
        ddm = rng.normal(dm_offset[0], dm_offset[1], block)
        gm = rng.normal(0.25, 0.05, block)
        ex = rng.standard_normal(block)
        lo, hi = max(first, b * block), min(first + n, (b + 1) * block)
        src, dst = slice(lo - b * block, hi - b * block), slice(lo - first, hi - first)
        inj[dst, 0], inj[dst, 1] = phi[src], dm0 + ddm[src]
        if flags[2]:
            inj[dst, 2] = gm[src]
        extra[dst] = ex[src]
    return inj, extra


def algorithmic_bytes_per_fit(C, B, s, n_share):
    """data + shared model + freqs/errs in, per-channel out."""
    return C * B * s + C * B * s / n_share + 40 * C + 512


# --------------------------------------------------------------------------
# CPU baseline workers (spawned BEFORE this process touches the GPU; they import
# NumPy/SciPy and the oracle only)
# --------------------------------------------------------------------------
def _cpu_fit(job):
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    path, model_path, x0, P, freqs, nu_fit, err, flags, log10_tau = job
    from oracle import pptoas_oracle as orc
    data = np.load(path)
    model = np.load(model_path)
    t0 = time.perf_counter()
    o = orc.fit_portrait_full(data, model, x0, P, freqs, [nu_fit] * 3, [None] * 3, err,
                              flags, log10_tau=log10_tau)
    return o.phi, o.DM, time.perf_counter() - t0


def _cpu_warm(_):
    from oracle import pptoas_oracle  # noqa: F401
    return os.getpid()


def start_cpu_pool():
    """One worker per physical core (psutil; half the logical cores otherwise),
    single-threaded BLAS/FFT in each."""
    import multiprocessing as mp
    try:
        import psutil
        workers = psutil.cpu_count(logical=False) or 0
    except ImportError:
        workers = 0
    avail = len(os.sched_getaffinity(0))
    if workers <= 0:
        workers = max(1, avail // 2)
    # (the fits are memory-bandwidth-bound: past ~64 concurrent workers the aggregate
    # rate of a 128-core host no longer grows -- 2.4 fits/s with 128, measured -- while the
    # leg's wall time does; 64 keeps the default run inside its time budget)
    workers = max(1, min(workers, avail, 64))
    # memory: ~0.7 GB per worker at 4096 x 2048 (measured); stay under 40 % of what
    # the host (or the container's cgroup) has free
    free_b = None
    try:
        import psutil
        free_b = psutil.virtual_memory().available
    except ImportError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
        if lim != "max":
            free_b = min(free_b, int(lim) - cur) if free_b is not None else int(lim) - cur
    except (OSError, ValueError):
        pass
    if free_b is not None:
        workers = max(1, min(workers, int(0.4 * free_b / 1.0e9)))
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, "1")
    pool = mp.get_context("spawn").Pool(workers)
    pool.map(_cpu_warm, range(workers), chunksize=1)     # imports done before the clock
    return pool, workers


# --------------------------------------------------------------------------
from pulseportraiture_amd.engine import EngineNotSupported  # noqa: E402  (no GPU touched by the import)


class Batch(object):
    """One workload's template, device-resident synthetic subints and guesses.
    eng: the Engine; args: the run's settings (seed, dm0, dm_offset, sigma, truth_guesses,
    measured_noise, method -- bench's argparse namespace or a stand-in); device: the
    torch device of the resident portraits."""

    def __init__(self, eng, args, device, workload, nsub, input_dtype, first_subint, seed_ns=0, reseed=False,
                 variant=None):
        """variant (SURVEY 8d's other regimes): dict(sigma=1.5) the low-S/N recipe of examples/example.py:28;
        dict(scint=True) per-channel scintillation gains (add_scintillation, pplib.py:1146-1174, nsin = 3,
        amax = 1, wmax = 5); dict(measured_noise=True) errs=None, the noise of every channel measured from
        the top quarter of its power spectrum as load_data always does (pplib.py:2227-2247);
        dict(mask_frac=0.2) an independent random channel mask per subint (zapped archives: the fit does
        C_i channels' work, pptoas.py:384-397)."""
        import torch
        v = dict(variant or {})
        self.sigma = float(v.get("sigma", args.sigma))
        self.scint = bool(v.get("scint", False))
        self.measured_noise = bool(v.get("measured_noise", getattr(args, "measured_noise", False)))
        self.mask_frac = float(v.get("mask_frac", 0.0))
        from pulseportraiture_amd import gmodel
        from pulseportraiture_amd.pplib import guess_fit_freq
        self.eng, self.args, self.device = eng, args, device
        self.reseed = reseed
        self.fused_unavailable = False
        self.workload = workload
        C, B, flags, log10_tau, nsub_def, note = WORKLOADS[workload]
        self.C, self.B, self.flags, self.log10_tau, self.note = C, B, flags, log10_tau, note
        self.nsub = nsub or nsub_def
        self.input_dtype, self.seed_ns = input_dtype, seed_ns
        self.s_bytes = 8 if input_dtype == "f64" else 4
        self.freqs, self.model, self.P0 = gmodel.example_model(C, B)
        self.nharm = self.eng.set_model(self.model)
        self.gen_slot, self.tau_rot = 0, 0.0
        if flags[3]:
            # scattered template: tau = 20 us at 1500 MHz, alpha = -4
            self.tau_rot = 20e-6 / self.P0
            taus = self.tau_rot * (self.freqs / 1500.0) ** -4.0
            k = np.arange(B // 2 + 1)
            smodel = np.fft.irfft(np.fft.rfft(self.model, axis=-1) /
                                  (1.0 + 2j * np.pi * np.outer(taus, k)), axis=-1)
            self.eng.set_model(smodel, slot=1)
            self.gen_slot = 1
        self.nu_fit = float(guess_fit_freq(self.freqs))
        # template profile of the 1-D seed fit: the mean profile, scattered with the
        # GUESSED tau at nu_fit when scattering is fitted (pptoas.py:430-452)
        self.seed_prof = self.model.mean(axis=0)
        if flags[3]:
            tg = 1.5 * self.tau_rot * (self.nu_fit / 1500.0) ** -4.0
            k = np.arange(B // 2 + 1)
            self.seed_prof = np.fft.irfft(np.fft.rfft(self.seed_prof) / (1.0 + 2j * np.pi * k * tg))
        self.data = torch.empty((self.nsub, C, B), device=self.device,
                                dtype=torch.float64 if input_dtype == "f64" else torch.float32)
        self.errs_dev = torch.full((self.nsub, C), self.sigma, dtype=torch.float64, device=self.device)
        self.P = np.full(self.nsub, self.P0)
        self.mask_host, self.mask_dev = None, None
        self.generate(first_subint)

    def generate(self, first_subint):
        """Fill the device buffer with subints [first, first + nsub) of the job
        (RNG keyed on the global subint index) and form their guesses."""
        nsub, flags = self.nsub, self.flags
        inj, unit = injected_params(self.args.seed, first_subint, nsub, flags, self.args.dm0, self.args.dm_offset)
        self.inj = inj
        gains = None
        if self.scint:
            # add_scintillation(random=True): sum of nsin sin^2 patterns across the band, per subint
            rng = np.random.default_rng([self.args.seed, 424242, first_subint])
            a, w, p = rng.uniform(0, 1.0, (nsub, 3)), rng.chisquare(5.0, (nsub, 3)), rng.uniform(0, 1, (nsub, 3))
            ramp = np.linspace(0.0, np.pi, self.C)
            gains = (a[:, :, None] * np.sin(w[:, :, None] * ramp[None, None, :] + p[:, :, None] * np.pi) ** 2).sum(axis=1)
        if self.mask_frac > 0.0:
            import torch
            rng = np.random.default_rng([self.args.seed, 535353, first_subint])
            self.mask_host = (rng.random((nsub, self.C)) >= self.mask_frac).astype(np.uint8)
            self.mask_dev = torch.from_numpy(self.mask_host).to(self.device)
        # mean frequency of the channels in use, per subint (pptoas.py:399)
        self.nu_mean = np.full(nsub, float(self.freqs.mean())) if self.mask_host is None else \
            (self.mask_host @ self.freqs) / self.mask_host.sum(axis=1)
        self.eng.synth_portraits(self.data, self.freqs, self.P, inj, self.sigma, self.args.seed,
                                 first_subint, slot=self.gen_slot, gains=gains)
        x0 = np.zeros((nsub, 5))
        x0[:, 1] = self.args.dm0
        if self.args.truth_guesses:
            phi_true = inj[:, 0] + DCONST * inj[:, 1] / self.P / self.nu_fit ** 2 + \
                DCONST ** 2 * inj[:, 2] / self.P / self.nu_fit ** 4
            x0[:, 0] = (phi_true + 1e-4 * unit + 0.5) % 1.0 - 0.5
            self.guess = "injected phase + 1e-4 rot noise"
        elif self.seed_ns > 0:
            self.guess = "device seed inside the timed fit (seed_ns=%d)" % self.seed_ns
        else:
            x0[:, 0] = self.pptoas_phase_guess()
            self.guess = "pptoas preamble (rotate to nu_mean, mean profile, fit_phase_shift Ns=100 with SciPy's simplex finish retraced)"
        if flags[3]:
            t0 = 1.5 * self.tau_rot * (self.nu_fit / 1500.0) ** -4.0
            x0[:, 3] = np.log10(t0) if self.log10_tau else t0
            x0[:, 4] = -4.0
        self.x0 = x0

    def pptoas_phase_guess(self):
        """pptoas.py:421-457 on the device: dedisperse every subint at the header
        DM to the mean frequency, average over channels, 1-D FFTFIT against the
        template's mean profile (Ns = 100 grid + SciPy's simplex finish, retraced:
        the guess the reference itself would start from), move the phase to nu_fit."""
        nu_mean = float(self.freqs.mean())
        w = np.ones((self.nsub, self.C)) if self.mask_host is None else self.mask_host.astype(np.float64)
        out = self.eng.reference_phase_seed(self.data, self.freqs, self.P, w,
                                            self.seed_prof, DM=np.full(self.nsub, self.args.dm0), nu_DM=nu_mean,
                                            Ns=100, finish='simplex')
        phi = out[:, 0] + DCONST * self.args.dm0 / self.P * (self.nu_fit ** -2 - nu_mean ** -2)
        return (phi + 0.5) % 1.0 - 0.5

    def fit(self, records=None, method=None, n=None):
        n = self.nsub if n is None else n         # (a ragged last sub-batch fits its first n)
        ref_seed = None
        if self.reseed and not getattr(self.args, "two_pass_seed", False) and not self.fused_unavailable:
            # the reference's own preamble inside the timed step, formed from the SAME pass over
            # the portraits as the fit (pp_seed_ref); batches without that path fall through
            numean = self.nu_mean[:n]
            ref_seed = dict(weights=None, model_profs=self.seed_prof, nu_mean=numean,
                            Ns=100, finish='simplex')
        elif self.reseed:
            # ... or with one more read of the portraits (rotation + channel mean +
            # fit_phase_shift), then the fit
            self.x0[:, 0] = self.pptoas_phase_guess()
        try:
            return self._fit(n, records, method, ref_seed)
        except EngineNotSupported:
            self.fused_unavailable = True
            return self.fit(records=records, method=method, n=n)

    def _fit(self, n, records, method, ref_seed, submit=False):
        eng = self.eng
        call = eng.enqueue if submit else eng.fit_batch
        return call(self.data[:n], self.freqs, self.P[:n], self.x0[:n],
                    errs=None if self.measured_noise else self.errs_dev[:n],
                    chan_mask=None if self.mask_dev is None else self.mask_dev[:n],
                    nu_fits=np.full((n, 3), self.nu_fit), fit_flags=self.flags,
                    log10_tau=self.log10_tau, per_channel="device",
                    seed_ns=self.seed_ns, method=method or self.args.method, records=records,
                    ref_seed=ref_seed)

    def can_pipeline(self):
        """Steps may be enqueued (pp_fit_enqueue) unless every step needs a synchronous call of its own
        first (the two-pass reference seed)."""
        return not (self.reseed and (getattr(self.args, "two_pass_seed", False) or self.fused_unavailable))

    def enqueue(self, records=None, method=None, n=None):
        """Queue this batch's fit on the engine's stream (pp_fit_enqueue) and return at once;
        eng.collect() returns the result.  (n: a ragged last sub-batch fits its first n.)"""
        n = self.nsub if n is None else n
        ref_seed = None
        if self.reseed:
            ref_seed = dict(weights=None, model_profs=self.seed_prof, nu_mean=self.nu_mean[:n], Ns=100, finish='simplex')
        self._fit(n, records, method, ref_seed, submit=True)

    def free(self):
        import torch
        del self.data, self.errs_dev
        self.mask_dev = None
        torch.cuda.empty_cache()


# --------------------------------------------------------------------------
def self_launch(ngpus, argv, script=None):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`
    in a CHILD process (never an exec, and before this process has imported torch or touched
    the GPU), hand its one JSON line on to stdout, everything else it writes there to stderr,
    and return its exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    print("bench.py: --gpus %d without RANK in the environment: starting %s" % (ngpus, " ".join(cmd)), file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd())
    nlines = 0
    for raw in proc.stdout:
        text = raw.decode("utf-8", "replace")
        if text.lstrip().startswith('{"metric"'):
            sys.stdout.write(text if text.endswith("\n") else text + "\n")
            sys.stdout.flush()
            nlines += 1
        else:
            sys.stderr.write(text)
    rc = proc.wait()
    if rc == 0 and nlines != 1:
        print("bench.py: the ranks printed %d JSON lines, expected 1" % nlines, file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="toa-4096x2048-phiDM",
                    choices=sorted(WORKLOADS))
    ap.add_argument("--nsub", type=int, default=0, help="subints per GPU per step")
    ap.add_argument("--total-nsub", type=int, default=0,
                    help="> 0: strong scaling -- this many subints in all, dealt to the ranks in "
                         "contiguous shards and fitted in device-generated sub-batches of --nsub")
    ap.add_argument("--input-dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--method", default="trust-ncg", choices=["trust-ncg", "newton"])
    ap.add_argument("--dm0", type=float, default=34.56789)
    ap.add_argument("--sigma", type=float, default=0.05)
    ap.add_argument("--dm-offset", type=float, nargs=2, default=[3e-4, 2e-4], metavar=("MEAN", "SIGMA"),
                    help="injected DM minus the guessed (header) DM: mean and scatter [pc cm^-3]")
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of every entry of other_workloads")
    ap.add_argument("--other-warmup", type=int, default=3, help="untimed steps in front of them")
    ap.add_argument("--cpu-configs", default="cfg2-512x1024-phiDM,cfg3-4096x2048-phiDMGM,cfg4-2048x2048-scat",
                    help="other_workloads entries that also get a CPU baseline (comma-separated; '' = none)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="subints the 1-core CPU leg fits (0 = as many as fit in ~10 s)")
    ap.add_argument("--seed-ns", type=int, default=0,
                    help="> 0: ignore the phase guesses and seed the phase on the device "
                         "with an N-point grid (the whole pptoas preamble + fit); "
                         "< 0: the reference's own preamble inside every timed step")
    ap.add_argument("--truth-guesses", action="store_true",
                    help="phase guesses = injected phase + 1e-4 rot of noise instead of the "
                         "fit_phase_shift seed (experiments)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="engine option (pp_set_option), e.g. scat_model=0; repeatable")
    ap.add_argument("--variant", default=None, choices=sorted(VARIANTS),
                    help="run the headline workload in one of SURVEY 8(d)'s other regimes (Batch.__init__)")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="2 / 3 = steps enqueued on the engine's stream (pp_fit_enqueue) before the oldest one is "
                         "collected: the host prepares a step while the previous ones run (3 when a step's solve and "
                         "post-fit stage ride in the next step's transform: option fuse_tail); 1 = synchronous calls")
    ap.add_argument("--measured-noise", action="store_true",
                    help="errs=None: the noise of every channel is measured from the top quarter of its "
                         "power spectrum inside the transform (get_noise_PS) instead of being given")
    ap.add_argument("--two-pass-seed", action="store_true",
                    help="--seed-ns -1: form the reference's guess in a pass of its own (pp_reference_phase_seed) "
                         "instead of inside the fit's single pass")
    ap.add_argument("--group", type=int, default=0,
                    help="--total-nsub: resident sub-batches per group (0 = up to three, as the free HBM allows)")
    ap.add_argument("--dump-records", default=None, metavar="PATH",
                    help="--total-nsub: rank 0 saves the gathered [total, 18] records there (.npy)")
    ap.add_argument("--harm-eps", type=float, default=None,
                    help="override the harmonic-truncation threshold (experiments)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # the plain command (what the driver runs for N = 1) asked for N > 1 ranks
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pool = None
    if world == 1 and not args.no_cpu_baseline:
        pool = start_cpu_pool()         # before anything initialises the GPU

    import torch
    import torch.distributed as dist
    from pulseportraiture_amd import dist as ppdist
    from pulseportraiture_amd import gmodel
    from pulseportraiture_amd.engine import Engine
    from pulseportraiture_amd.pplib import guess_fit_freq

    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d" % (args.gpus, world))
    # PP_BENCH_SHARE_GPU=1 (tests on a one-GPU box): the ranks share the GPUs there are and talk
    # over gloo -- the whole N > 1 path (shards, barriers, max over ranks, the gather) with the
    # real engine, minus RCCL, which wants one device per rank
    share = os.environ.get("PP_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share else device      # where the small collectives' tensors live
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # launched by torchrun
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes a version banner to stdout when its first communicator comes
        # up; stdout is reserved for the one JSON line, so route fd 1 to stderr
        # until the communicator exists
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if share:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    eng = Engine(local_rank)
    if args.harm_eps is not None:
        eng.set_option("harm_eps", args.harm_eps)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        eng.set_option(name, float(val))

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(batch, steps, warmup, method=None):
        """`steps` passes over the resident batch; records of all steps stay on the
        device and are gathered once before the clock stops.  With --pipeline 2 (default) step
        k + 1 is enqueued on the engine's stream (pp_fit_enqueue) before step k is collected:
        while one step's kernels run the host marshals and queues the next, so the GPU never
        waits for the host between steps.  One stream: kernels do not overlap, the per-kernel
        HIP-event times stay exact; every step is still a whole fit of the whole batch."""
        recs = torch.zeros((steps, batch.nsub, ppdist.RECORD_WIDTH), dtype=torch.float64, device=device)
        # (the interpreter's cyclic garbage collector is held off over the timed steps, as timeit does: with torch
        # and NumPy loaded a full collection takes 30-40 ms and used to land in one timed step or another.  Collected
        # HERE, before the warm-up steps: a device left idle for those 40 ms starts the timed region at a ramping
        # clock, 1-3 ms longer for its first step -- profiles/r04_strong_gap.txt)
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        for _ in range(warmup):
            batch.fit(method=method)
        piped = args.pipeline > 1 and batch.can_pipeline()

        def run(nsteps, out):
            res = None
            if piped:
                trace = [] if os.environ.get("PP_BENCH_STEP_TIMES") else None
                depth = max(2, min(3, args.pipeline))       # enqueued steps in flight
                for k in range(nsteps):
                    t_a = time.perf_counter()
                    batch.enqueue(records=None if out is None else out[k], method=method)
                    t_b = time.perf_counter()
                    if k >= depth - 1:
                        res = eng.collect()
                    if trace is not None:
                        trace.append((1e3 * (t_b - t_a), 1e3 * (time.perf_counter() - t_b), torch.cuda.memory_reserved() / 2 ** 20))
                for _ in range(min(depth - 1, nsteps)):
                    res = eng.collect()
                if trace:
                    print("step times (enqueue, collect ms; torch MiB reserved): " + " ".join("%.2f/%.2f/%.0f" % t for t in trace), file=sys.stderr)
            else:
                for k in range(nsteps):
                    res = batch.fit(records=None if out is None else out[k], method=method)
            return res

        if piped and warmup > 0:
            # (three untimed steps in the timed loop's own pattern: up to three steps' output tensors are
            # alive at once there -- two pending and the last result -- and the first time torch's caching
            # allocator has to grow for that, hipMalloc waits for the device: a ~20 ms stall that landed in
            # one timed 3-step workload or another)
            run(4, None)
        eng.set_option("profile", 1)
        eng.kernel_times(reset=True)
        fence()
        t0 = time.perf_counter()
        res = run(steps, recs)
        gathered = ppdist.gather_records(recs.view(-1, ppdist.RECORD_WIDTH))   # one RCCL gather
        fence()
        elapsed = time.perf_counter() - t0
        if gc_was_on:
            gc.enable()
        eng.set_option("profile", 0)
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        timed.piped = piped
        return res, gathered, elapsed, eng.kernel_times()

    def summary(batch, res, elapsed, ktimes, steps, nfits):
        fam = max((k for k in ktimes if ktimes[k][1] > 0), key=lambda k: ktimes[k][0])
        per_step_s = ktimes[fam][0] / steps
        abytes = algorithmic_bytes_per_fit(batch.C, batch.B, batch.s_bytes, batch.nsub)
        if getattr(batch, "mask_host", None) is not None:
            # the fit of a zapped subint does C_i channels' work (pptoas.py:384-397): the data term counts
            # the channels in use, the shared template and the per-channel arrays all of them
            c_eff = float(batch.mask_host.sum()) / batch.nsub
            abytes += (c_eff - batch.C) * batch.B * batch.s_bytes
        achieved = abytes * batch.nsub / per_step_s / 1e9
        return fam, per_step_s, abytes, achieved, {
            "fits_per_s": round(nfits / elapsed, 2),
            "ms_per_step": round(1e3 * elapsed / steps, 3),
            "dominant_kernel": fam,
            "hbm_frac_of_8TBps": round(achieved / HBM_PEAK_GBPS, 4),
            "kernels_ms_per_step": {k: round(1e3 * v[0] / steps, 4) for k, v in ktimes.items() if v[1] > 0},
            "nfeval_mean": float(np.mean(res["nfeval"])), "nfeval_max": int(np.max(res["nfeval"])),
            "npass_mean": float(np.mean(res["npass"])), "npass_max": int(np.max(res["npass"])),
            "left_one_pass_flow": int(np.sum(res["npass"] > 1)),
            "return_codes": {str(k): int(v) for k, v in zip(*np.unique(res["return_code"],
                                                                      return_counts=True))}}

    # ======================================================================
    def sync():
        eng.synchronize()
        torch.cuda.synchronize()

    def make_batch(workload, nsub, first):
        return Batch(eng, args, device, workload, nsub, args.input_dtype, first, seed_ns=args.seed_ns)

    if args.total_nsub > 0:
        sline = strong_scaling(args, make_batch, sync, fence, device, rank, world, use_dist,
                               collect=eng.collect if args.pipeline > 1 else None)
        if rank == 0:
            print(json.dumps(sline))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    batch = Batch(eng, args, device, args.workload, args.nsub, args.input_dtype,
                  rank * (args.nsub or WORKLOADS[args.workload][4]),
                  seed_ns=max(args.seed_ns, 0), reseed=(args.seed_ns < 0),
                  variant=VARIANTS[args.variant] if args.variant else None)
    res, gathered, elapsed, ktimes = timed(batch, args.steps, args.warmup)

    line = None
    if rank == 0:
        nsub, C, B = batch.nsub, batch.C, batch.B
        flags_scat = bool(batch.flags[3] or batch.flags[4])
        fam, per_step_s, abytes, achieved, summ = summary(batch, res, elapsed, ktimes, args.steps,
                                                          nsub * world * args.steps)
        # HBM bytes per launch of that kernel from the PMC passes (FETCH_SIZE x2 +
        # WRITE_SIZE, collected separately with rocprofv3 --pmc and committed under
        # profiles/); null when no matching profile exists
        traffic, co_limit, traffic_source = None, None, None
        try:
            tp = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if (tp["workload"] == args.workload and tp["input_dtype"] == args.input_dtype
                    and tp["kernel"] == fam):
                traffic = tp["hbm_bytes_per_fit"] * nsub
                # (a constant of the committed PMC passes -- rocprofv3 --pmc cannot run inside this process --, not
                # a measurement of THIS run; tools/profile_round.sh regenerates it)
                traffic_source = "profiles/traffic_latest.json <- " + str(tp.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile_round.sh"))
                if "valu_issue_frac_per_wave" in tp:
                    co_limit = {"resource": "power cap (shader clock under this kernel against 2.4 GHz) "
                                            "with the f64 VALU issuing busy_frac of the time",
                                "busy_frac": round(tp["valu_issue_frac_per_wave"] *
                                                   tp.get("waves_per_simd", 1), 3),
                                "shader_clock_ghz": tp.get("shader_clock_ghz"),
                                "source": tp.get("counters_source")}
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": fam, "co_limit": co_limit,
                    "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": abytes * nsub,
                    "algorithmic_bytes_per_fit": abytes,
                    "fits_per_launch_group": nsub,
                    "carries": ("each launch also works off the PREVIOUS step's solve + post-fit stage as tickets between its "
                                "rows (engine option fuse_tail; the last timed step's go out by the stand-alone kernels): "
                                "its duration is the whole fit's, transform + solve + post-fit stage"
                                if (getattr(timed, "piped", False) and eng.get_option("fuse_tail") > 0 and B == 2048
                                    and not flags_scat and args.seed_ns >= 0) else None),
                    "launches_per_step": ktimes[fam][1] / args.steps,
                    "ms_per_step_in_kernel": round(1e3 * per_step_s, 4),
                    "all_kernels_ms_per_step": summ["kernels_ms_per_step"]}
        rec = gathered.cpu().numpy() if hasattr(gathered, "cpu") else np.asarray(gathered)
        line = {"metric": "subint_fits_per_sec", "value": summ["fits_per_s"],
                "unit": "fits/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": summ["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": args.workload, "note": batch.note,
                           "nsub_per_gpu_per_step": nsub, "nchan": C, "nbin": B,
                           "fit_flags": batch.flags, "input_dtype": args.input_dtype,
                           "bytes_per_sample_resident": batch.s_bytes, "dm0": args.dm0, "dm_offset": list(args.dm_offset),
                           "sigma": batch.sigma, "variant": args.variant, "model_harmonics_kept": batch.nharm,
                           "method": args.method, "phase_guesses": batch.guess,
                           "device_phase_seed_ns": args.seed_ns,
                           "steps_in_flight": max(2, min(3, args.pipeline)) if getattr(timed, "piped", False) else 1,
                           "parallelism": "subint shards, %d rank(s), records kept in HBM, "
                                          "1 gather at the end" % world},
                "roofline": roofline,
                "convergence": {"nfeval_mean": summ["nfeval_mean"], "nfeval_max": summ["nfeval_max"],
                                "npass_mean": summ["npass_mean"], "npass_max": summ["npass_max"],
                                "note": "nfeval: objective evaluations as the reference counts them (SciPy's "
                                        "nfev); npass: how many of them were passes over the data",
                                "return_codes": summ["return_codes"]},
                "gathered_records": {"rows": int(rec.shape[0]),
                                     "checksum": ppdist.records_checksum(rec)["column_sums"][:3]}}
    # ---- the other workloads, same process, --other-steps timed steps each (N = 1 only) ----
    if world == 1 and not args.no_other_workloads:
        cpu_cases = {}
        if pool is not None:
            cpu_cases["headline"] = cpu_case(batch, res, 64)
        cpu_keys = [k for k in args.cpu_configs.split(",") if k]
        batch.free()
        others = {}
        plan = [("seeded", args.workload, args.input_dtype, 100, None),
                ("reference_seed_in_step", args.workload, args.input_dtype, -1, None),
                ("f32", args.workload, "f32", 0, None),
                # SURVEY 8(d)'s other regimes of the headline shape (Batch.__init__)
                ("lowsnr_sigma1.5", args.workload, args.input_dtype, 0, None, VARIANTS["lowsnr_sigma1.5"]),
                ("scint", args.workload, args.input_dtype, 0, None, VARIANTS["scint"]),
                ("measured_noise", args.workload, args.input_dtype, 0, None, VARIANTS["measured_noise"]),
                ("masked20", args.workload, args.input_dtype, 0, None, VARIANTS["masked20"]),
                ("masked20_reference_seed_in_step", args.workload, args.input_dtype, -1, None, VARIANTS["masked20"]),
                # a template that keeps every harmonic (data-derived spline / PCA templates do: harm_eps = 0)
                ("full_spectrum_template", args.workload, "f64", 0, "full"),
                ("cfg2-512x1024-phiDM", "cfg2-512x1024-phiDM", "f64", 0, None),
                ("cfg3-4096x2048-phiDMGM", "cfg3-4096x2048-phiDMGM", "f64", 0, None),
                ("cfg4-2048x2048-scat", "cfg4-2048x2048-scat", "f64", 0, None),
                ("cfg4-2048x2048-scat-newton", "cfg4-2048x2048-scat", "f64", 0, "newton"),
                ("cfg4_reference_seed_in_step", "cfg4-2048x2048-scat", "f64", -1, None)]
        for entry in plan:
            key, wl, dt, sns, meth = entry[:5]
            variant = entry[5] if len(entry) > 5 else None
            if wl == args.workload and dt == args.input_dtype and sns == args.seed_ns and meth is None and not variant:
                continue
            full = (meth == "full")
            if full:
                meth = None
                if args.harm_eps is not None:
                    continue
                eng.set_option("harm_eps", 0.0)
            try:
                b = Batch(eng, args, device, wl, 0, dt, 0, seed_ns=max(sns, 0), reseed=(sns < 0), variant=variant)
                if sns < 0:
                    b.guess = "the reference's preamble INSIDE the timed step (rotation + channel mean + fit_phase_shift " \
                              "with the simplex finish, from the fit's own single pass over the portraits), then " \
                              "trust-ncg from that guess"
                r, _, el, kt = timed(b, args.other_steps, args.other_warmup, method=meth)
                _, _, _, _, sm = summary(b, r, el, kt, args.other_steps, b.nsub * args.other_steps)
                sm["steps"], sm["warmup"] = args.other_steps, args.other_warmup
                if pool is not None and key in cpu_keys:
                    cpu_cases[key] = cpu_case(b, r, 16)
                sm["steps_in_flight"] = max(2, min(3, args.pipeline)) if getattr(timed, "piped", False) else 1
                sm.update(workload=wl, input_dtype=dt, seed_ns=max(sns, 0), nsub=b.nsub,
                          method=meth or args.method, phase_guesses=b.guess)
                # recovered values against the injected ones, in units of the errors
                sm["max_abs_dDM_over_err"] = float(np.max(np.abs(r["params"][:, 1] - b.inj[:, 1]) /
                                                          r["param_errs"][:, 1]))
                if full:
                    sm["model_harmonics_kept"] = b.nharm
                if variant:
                    sm["variant"] = variant
                    if b.mask_host is not None:
                        sm["mean_channels_in_use"] = float(b.mask_host.sum()) / b.nsub
                if sns < 0:
                    sm["single_pass"] = not b.fused_unavailable
                others[key] = sm
                b.free()
            except Exception as exc:      # a secondary workload must not lose the headline
                others[key] = {"error": repr(exc)}
            finally:
                if full:
                    eng.set_option("harm_eps", 2.0 ** -50)
        # configs[4]'s flow (contiguous shard, device-generated sub-batches, ragged last one, one
        # gather) at a size one GPU finishes in seconds
        try:
            import copy
            sargs = copy.copy(args)
            sargs.total_nsub, sargs.nsub, sargs.dump_records = 3000, 1024, None
            sl = strong_scaling(sargs, make_batch, sync, fence, device, rank, world, use_dist,
                                collect=eng.collect if args.pipeline > 1 else None)
            others["strong_3000"] = {"fits_per_s": sl["value"], "ms_total": sl["ms_per_step"],
                                     "gather_ms": sl["gather_ms"], "wall_s_all_in": sl["wall_s"],
                                     "fits_per_s_all_in": sl["fits_per_s_all_in"], "rows": sl["gathered_records"]["rows"],
                                     "sub_batches": sl["config"]["sub_batches_rank0"],
                                     "resident_sub_batches": sl["config"]["resident_sub_batches"],
                                     "steps_in_flight": sl["config"]["steps_in_flight"],
                                     "max_abs_dDM_over_err": sl["max_abs_dDM_over_err"],
                                     "workload": args.workload, "scaling": "strong"}
        except Exception as exc:
            others["strong_3000"] = {"error": repr(exc)}
        line["other_workloads"] = others
        if pool is not None:
            line["cpu_baseline"] = cpu_baseline(pool, cpu_cases, args.cpu_sample)
    elif pool is not None and rank == 0:
        line["cpu_baseline"] = cpu_baseline(pool, {"headline": cpu_case(batch, res, 64)}, args.cpu_sample)
    if pool is not None:
        pool[0].close()
        pool[0].join()
    if rank == 0:
        # the LAST key: everything this run measured, in fits/s (a reader that keeps only the tail of the line
        # still gets every workload)
        fps = {"headline:" + args.workload: line["value"]}
        for k, v in (line.get("other_workloads") or {}).items():
            fps[k] = v.get("fits_per_s") if isinstance(v, dict) else None
        cb = line.get("cpu_baseline")
        if cb:
            fps["cpu_pool:headline"] = cb["value"]
            fps["cpu_1core:headline"] = cb["one_core"]["value"]
            for k, v in (cb.get("per_config") or {}).items():
                fps["cpu_pool:" + k] = v.get("value")
                fps["cpu_1core:" + k] = (v.get("one_core") or {}).get("value")
        line["fits_per_s_summary"] = fps
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def strong_scaling(args, make_batch, sync, fence, device, rank, world, use_dist, collect=None):
    """configs[4] as written: --total-nsub subints in all, rank r owns the contiguous
    shard shard_range(total, r, world) and fits it in device-generated sub-batches
    of --nsub; every record stays in HBM and ONE gather at the end brings them to rank 0.

    The shard is worked through in GROUPS of up to three resident sub-batches (as many
    as the free HBM holds: 3 x 68.7 GB at the headline shape): a group is generated, one
    untimed fit brings the device to the state the timed fits run in (the generator and
    the idle host work around it leave the shader clock ramping: a fit that follows them
    runs 1.3-2.7 ms longer on the device than the same fit repeated, profiles/r04_strong_gap.txt
    -- the same reason `--warmup` steps exist), then the clock runs while the group's fits
    are ENQUEUED BACK TO BACK (pp_fit_enqueue three deep, as the weak loop does): inputs are
    resident when their timed region starts, the device never idles inside it.
    Returns the JSON line's dict on rank 0 (None elsewhere).  `make_batch(workload, nsub,
    first)` builds a resident batch (bench's Batch; a stub in the CPU tests), `sync()`
    drains the device, `collect()` returns the oldest enqueued batch's result (Engine.collect)."""
    import torch
    import torch.distributed as dist
    from pulseportraiture_amd import dist as ppdist
    C, B, flags, log10_tau, nsub_def, note = WORKLOADS[args.workload]
    nsub = args.nsub or nsub_def
    lo, hi = ppdist.shard_range(args.total_nsub, rank, world)
    counts = [b - a for a, b in (ppdist.shard_range(args.total_nsub, r, world) for r in range(world))]
    fence()
    t_wall0 = time.perf_counter()               # all-in clock: generation, warm-up and ramp fits, the fits, the gather
    recs = torch.zeros((hi - lo, ppdist.RECORD_WIDTH), dtype=torch.float64, device=device)
    nbatches = max(1, -(-(hi - lo) // nsub))
    group = max(1, min(3, nbatches, int(getattr(args, "group", 0) or 3)))
    if str(device) != "cpu" and torch.cuda.is_available():
        free_b, _ = torch.cuda.mem_get_info(device)
        per = nsub * C * B * (8 if args.input_dtype == "f64" else 4) + nsub * C * 8
        group = max(1, min(group, int(0.8 * free_b // per)))
    batches = [make_batch(args.workload, nsub, lo + g * nsub) for g in range(group)]
    piped = collect is not None and all(hasattr(b, "enqueue") and getattr(b, "can_pipeline", lambda: False)() for b in batches)
    batches[0].fit()                             # warm-up (untimed)
    if piped:
        # (... and three fits in the timed loop's own pattern: every set of staging / work buffers of
        # pp_fit_enqueue -- three since the fused tail -- is allocated on first use)
        for _ in range(3):
            batches[0].enqueue()
        for _ in range(3):
            collect()
    gc.collect()
    gc.disable()                                 # (a full collection takes 30-40 ms with torch loaded: not inside a timed fit)
    fit_s, done = 0.0, 0
    worst = 0.0
    sub_batches = []
    first_group = True
    while done < hi - lo:
        todo = []                                # (batch, offset in the shard, subints)
        for g in range(group):
            off = done + g * nsub
            if off >= hi - lo:
                break
            if not first_group:
                batches[g].generate(lo + off)
            todo.append((batches[g], off, min(nsub, hi - lo - off)))
        first_group = False
        todo[0][0].fit(n=todo[0][2])             # untimed: the clock ramp after the generator
        sync()
        t0 = time.perf_counter()
        results = []
        if piped:
            depth = 3        # (a batch's solve and post-fit stage ride in the next batch's transform: engine option fuse_tail)
            for k, (b, off, n) in enumerate(todo):
                b.enqueue(records=recs[off:off + n], n=n)
                if k >= depth - 1:
                    results.append(collect())
            while len(results) < len(todo):
                results.append(collect())
        else:
            for b, off, n in todo:
                results.append(b.fit(records=recs[off:off + n], n=n))
        sync()
        fit_s += time.perf_counter() - t0
        for (b, off, n), res in zip(todo, results):
            worst = max(worst, float(np.max(np.abs(res["params"][:n, 1] - b.inj[:n, 1]) / res["param_errs"][:n, 1])))
            sub_batches.append(n)
            done += n
    fence()
    t0 = time.perf_counter()
    gathered = ppdist.gather_records(recs, counts=counts)
    fence()
    gather_s = time.perf_counter() - t0
    wall_s = time.perf_counter() - t_wall0
    gc.enable()
    total_s = fit_s + gather_s
    if use_dist:
        t = torch.tensor([total_s, worst, wall_s], dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total_s, worst, wall_s = float(t[0].item()), float(t[1].item()), float(t[2].item())
    guess = batches[0].guess
    for b in batches:
        if hasattr(b, "free"):
            b.free()
    if rank != 0:
        return None
    rec = gathered.cpu().numpy() if hasattr(gathered, "cpu") else np.asarray(gathered)
    if getattr(args, "dump_records", None):
        np.save(args.dump_records, rec)
    cs = ppdist.records_checksum(rec)
    return {
        "metric": "subint_fits_per_sec", "value": round(args.total_nsub / total_s, 2),
        "unit": "fits/s", "n_gpus": world, "steps": 1, "warmup": 1,
        "ms_per_step": round(1e3 * total_s, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "note": note, "total_nsub": args.total_nsub,
                   "fits_per_rank": counts, "sub_batch": nsub, "sub_batches_rank0": sub_batches,
                   "resident_sub_batches": group, "steps_in_flight": 3 if piped else 1,
                   "nchan": C, "nbin": B,
                   "fit_flags": flags, "input_dtype": args.input_dtype, "method": args.method,
                   "phase_guesses": guess,
                   "timed": "per group of %d resident sub-batch(es): generation and ONE untimed fit (clock ramp) outside "
                            "the clock, then the group's fits enqueued back to back inside it; + the one gather of all "
                            "records" % group,
                   "parallelism": "contiguous subint shards over %d rank(s), records kept in "
                                  "HBM, 1 gather at the end" % world},
        "gather_ms": round(1e3 * gather_s, 3),
        # everything this job did between its first and last barrier, max over ranks: batch construction and
        # generation of every sub-batch, the warm-up fits, one untimed ramp fit per group, the timed fits, the gather
        "wall_s": round(wall_s, 4), "fits_per_s_all_in": round(args.total_nsub / wall_s, 2),
        "gathered_records": {"rows": cs["rows"], "checksum": cs["column_sums"][:3],
                             "column_sums": cs["column_sums"],
                             "return_code_sum": cs["column_sums"][17]},
        "max_abs_dDM_over_err": worst}


def cpu_case(batch, res, n):
    """Host copies of the first n subints of a fitted batch: what the CPU legs need."""
    n = min(n, batch.nsub)
    return dict(data=batch.data[:n].cpu().numpy(), model=batch.model, freqs=batch.freqs, P=batch.P[:n].copy(),
                x0=batch.x0[:n].copy(), sigma=batch.sigma, nu_fit=batch.nu_fit, flags=batch.flags,
                log10_tau=batch.log10_tau, params=np.asarray(res["params"])[:n].copy(),
                shape="%dx%d" % (batch.C, batch.B))


def _cpu_one_core(job_list):
    """In a pool worker: the jobs one after the other on this worker's core; seconds per fit."""
    out = []
    for job in job_list:
        out.append(_cpu_fit(job))
    return out


def cpu_baseline(pool, cases, nsample):
    """The CPU oracle (a NumPy/SciPy restatement of the reference algorithm) on bounded samples of the
    very batches the GPU fitted.  Per case (the headline shape and configs[1..3]): (i) one
    process, one core; (ii) a pool with one single-threaded worker per physical core, every worker one
    subint -- 64 fits for configs[1], max(8, workers) for the 2048-bin shapes.  The one-core legs of the
    other configs run in three pool workers WHILE this process times the headline's (four busy cores of
    the host: no contention to speak of).  Parity of the GPU answers on the distinct subints."""
    import tempfile
    from oracle import pptoas_oracle as orc
    pool, workers = pool
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    out_cfg = {}
    with tempfile.TemporaryDirectory(dir=shm) as tmp:
        # every case's distinct subints and template on the RAM disk
        files = {}
        for name, cs in cases.items():
            nd = min(16, cs["data"].shape[0])
            mpath = os.path.join(tmp, "%s_model.npy" % name)
            np.save(mpath, cs["model"])
            paths = []
            for i in range(nd):
                paths.append(os.path.join(tmp, "%s_sub%d.npy" % (name, i)))
                np.save(paths[-1], cs["data"][i].astype(np.float64))
            files[name] = (mpath, paths)

        def job(name, j):
            cs = cases[name]
            mpath, paths = files[name]
            i = j % len(paths)
            return (paths[i], mpath, cs["x0"][i], cs["P"][i], cs["freqs"], cs["nu_fit"],
                    np.full(len(cs["freqs"]), cs["sigma"]), cs["flags"], cs["log10_tau"])

        def parity(name, fits):
            cs = cases[name]
            nd = len(files[name][1])
            dphi = max(abs(((phi - cs["params"][j % nd, 0]) + 0.5) % 1.0 - 0.5) for j, (phi, DM, _) in fits)
            dDM = max(abs(DM - cs["params"][j % nd, 1]) for j, (phi, DM, _) in fits)
            return dphi, dDM

        others = [k for k in cases if k != "headline"]
        # (i) one core: the other configs' legs start now, in a worker each (about 10 s of fits: one fit of a
        # 2048-bin shape takes 20-25 s, configs[1] 1 s)
        n_one = {k: (8 if cases[k]["data"].shape[1] * cases[k]["data"].shape[2] <= 1 << 20 else 1) for k in others}
        pending = {k: pool.apply_async(_cpu_one_core, ([job(k, j) for j in range(n_one[k])],)) for k in others}
        head = None
        if "headline" in cases:
            cs = cases["headline"]
            errs = np.full(len(cs["freqs"]), cs["sigma"])
            budget_s, cap = 10.0, min(32, cs["data"].shape[0])
            want = cap if nsample <= 0 else max(1, min(nsample, cs["data"].shape[0]))
            t0 = time.perf_counter()
            outs = []
            for i in range(want):
                o = orc.fit_portrait_full(cs["data"][i].astype(np.float64), cs["model"], cs["x0"][i], cs["P"][i], cs["freqs"],
                                          [cs["nu_fit"]] * 3, [None] * 3, errs, cs["flags"], log10_tau=cs["log10_tau"])
                outs.append((i, (o.phi, o.DM, 0.0)))
                el = time.perf_counter() - t0
                if nsample <= 0 and el + el / (i + 1) > budget_s:
                    break
            n1, dt1 = len(outs), time.perf_counter() - t0
            head = dict(n1=n1, dt1=dt1, par=parity("headline", outs))
        one = {}
        for k in others:
            fits = pending[k].get()
            secs = float(sum(f[2] for f in fits))
            one[k] = dict(n=len(fits), secs=secs, par=parity(k, list(enumerate(fits))))
        # (ii) the pools, one case after the other
        def pooled(name, njobs, nrounds):
            jobs = [job(name, j) for j in range(njobs)]
            rounds, pres = [], None
            for _ in range(nrounds):
                t0 = time.perf_counter()
                pres = pool.map(_cpu_fit, jobs, chunksize=1)
                rounds.append(time.perf_counter() - t0)
            return rounds, pres
        result = None
        if head is not None:
            njobs = max(8, workers)
            # two rounds of the same jobs: the spread says how repeatable the rate is (the better round is the
            # baseline: the first one also pays for cold page caches and whatever else the host was doing --
            # 63 s against 17 s on one box; both are listed)
            rounds, pres = pooled("headline", njobs, 2)
            dtp = float(min(rounds))
            dphi, dDM = parity("headline", list(enumerate(pres)))
            dphi, dDM = max(dphi, head["par"][0]), max(dDM, head["par"][1])
            nd = len(files["headline"][1])
            result = {"value": round(njobs / dtp, 4), "unit": "fits/s", "cores": workers, "kind": "port",
                      "workers": workers, "host_cpu_count": os.cpu_count(),
                      "sample": "%d round(s) of %d fits (%d distinct subints of the timed batch, one per worker, "
                                "single-threaded NumPy/SciPy each), whole fit_portrait_full "
                                "(oracle/pptoas_oracle.py), %s s wall; mean %.1f s per fit inside a worker"
                                % (len(rounds), njobs, nd, " and ".join("%.1f" % r for r in rounds),
                                   float(np.mean([p[2] for p in pres]))),
                      "rounds_fits_per_s": [round(njobs / r, 4) for r in rounds],
                      "one_core": {"value": round(head["n1"] / head["dt1"], 5), "unit": "fits/s", "cores": 1,
                                   "sample": "%d subint(s), %.1f s" % (head["n1"], head["dt1"])},
                      "parity_on_sample": {"max_abs_dphi": dphi, "max_abs_dDM": dDM, "subints": max(head["n1"], nd)}}
        for k in others:
            small = cases[k]["data"].shape[1] * cases[k]["data"].shape[2] <= 1 << 20
            njobs = 64 if small else max(8, workers)
            rounds, pres = pooled(k, njobs, 1)
            dphi, dDM = parity(k, list(enumerate(pres)))
            dphi, dDM = max(dphi, one[k]["par"][0]), max(dDM, one[k]["par"][1])
            out_cfg[k] = {"value": round(njobs / rounds[0], 4), "unit": "fits/s", "cores": workers, "kind": "port",
                          "shape": cases[k]["shape"], "fit_flags": cases[k]["flags"],
                          "sample": "%d fits (%d distinct subints of the timed batch) on %d single-threaded workers, %.1f s wall; "
                                    "mean %.1f s per fit inside a worker" % (njobs, len(files[k][1]), workers, rounds[0],
                                                                             float(np.mean([p[2] for p in pres]))),
                          "one_core": {"value": round(one[k]["n"] / one[k]["secs"], 5), "unit": "fits/s", "cores": 1,
                                       "sample": "%d subint(s), %.1f s, in a pool worker of its own while the headline's "
                                                 "one-core leg ran" % (one[k]["n"], one[k]["secs"])},
                          "parity_on_sample": {"max_abs_dphi": dphi, "max_abs_dDM": dDM}}
    if result is None:
        result = {"value": None, "unit": "fits/s", "cores": workers, "kind": "port"}
    result["per_config"] = out_cfg
    return result


if __name__ == "__main__":
    main()
