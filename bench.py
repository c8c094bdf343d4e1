#!/usr/bin/env python3
"""Benchmark of the wideband-TOA hot path: subint fits/s on synthetic portraits
resident in HBM (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the whole fit (rFFT, cross-spectrum, trust-region
solve, zero-covariance frequency, errors, S/N, chi2) over one batch of
`nsub` subints per GPU; the batch is generated on the device before the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
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
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s copy)
DCONST = 0.000241 ** -1


def algorithmic_bytes_per_fit(C, B, s, n_share):
    """data + shared model + freqs/errs in, per-channel out."""
    return C * B * s + C * B * s / n_share + 40 * C + 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="toa-4096x2048-phiDM",
                    choices=sorted(WORKLOADS))
    ap.add_argument("--nsub", type=int, default=0, help="subints per GPU per step")
    ap.add_argument("--input-dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--dm0", type=float, default=34.56789)
    ap.add_argument("--sigma", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="subints the CPU baseline fits (0 = as many as fit in ~12 s, at most 64)")
    ap.add_argument("--seed-ns", type=int, default=0,
                    help="> 0: ignore the phase guesses and seed the phase on the device "
                         "with an N-point grid (the whole pptoas preamble + fit)")
    ap.add_argument("--harm-eps", type=float, default=None,
                    help="override the harmonic-truncation threshold (experiments)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pulseportraiture_amd import dist as ppdist
    from pulseportraiture_amd import gmodel
    from pulseportraiture_amd.engine import Engine
    from pulseportraiture_amd.pplib import guess_fit_freq

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d"
                             % args.gpus)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
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
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=device)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    C, B, flags, log10_tau, nsub_def, note = WORKLOADS[args.workload]
    nsub = args.nsub or nsub_def
    tdtype = torch.float64 if args.input_dtype == "f64" else torch.float32
    s_bytes = 8 if args.input_dtype == "f64" else 4

    eng = Engine(local_rank)
    if args.harm_eps is not None:
        eng.set_option("harm_eps", args.harm_eps)
    freqs, model, P0 = gmodel.example_model(C, B)
    nharm = eng.set_model(model)
    # ---- synthetic batch, generated on the device (weak scaling: every rank
    # owns nsub subints; global subint index keys the RNG) ----
    rng = np.random.default_rng([args.seed, rank])
    first = rank * nsub
    P = np.full(nsub, P0)
    inj = np.zeros((nsub, 3))
    inj[:, 0] = rng.uniform(-0.5, 0.5, nsub)
    inj[:, 1] = args.dm0 + rng.normal(3e-4, 2e-4, nsub)
    tau_rot = 0.0
    if flags[2]:
        inj[:, 2] = rng.normal(0.25, 0.05, nsub)
    data = torch.empty((nsub, C, B), dtype=tdtype, device=device)
    if flags[3]:
        # scattered template: tau = 20 us at 1500 MHz, alpha = -4
        tau_rot = 20e-6 / P0
        taus = tau_rot * (freqs / 1500.0) ** -4.0
        k = np.arange(B // 2 + 1)
        smodel = np.fft.irfft(np.fft.rfft(model, axis=-1) /
                              (1.0 + 2j * np.pi * np.outer(taus, k)), axis=-1)
        eng.set_model(smodel, slot=1)
        eng.synth_portraits(data, freqs, P, inj, args.sigma, args.seed, first, slot=1)
    else:
        eng.synth_portraits(data, freqs, P, inj, args.sigma, args.seed, first)
    # ---- initial guesses as the caller forms them (pptoas.py:399-460): DM =
    # header DM, phase from the 1-D seed fit (good to ~1e-4 rot), at nu_fit ----
    nu_fit = float(guess_fit_freq(freqs))
    x0 = np.zeros((nsub, 5))
    phi_true = inj[:, 0] + DCONST * inj[:, 1] / P / nu_fit ** 2 + \
        DCONST ** 2 * inj[:, 2] / P / nu_fit ** 4
    x0[:, 0] = (phi_true + 1e-4 * rng.standard_normal(nsub) + 0.5) % 1.0 - 0.5
    x0[:, 1] = args.dm0
    if flags[3]:
        x0[:, 3] = np.log10(1.5 * tau_rot * (nu_fit / 1500.0) ** -4.0) \
            if log10_tau else 1.5 * tau_rot * (nu_fit / 1500.0) ** -4.0
        x0[:, 4] = -4.0
    errs = np.full((nsub, C), args.sigma)
    errs_dev = torch.full((nsub, C), args.sigma, dtype=torch.float64, device=device)
    nu_fits = np.full((nsub, 3), nu_fit)
    # per-channel inputs and outputs stay in HBM (inputs resident before the timed
    # region; the fitted TOA records are what leaves the GPU)
    kw = dict(errs=errs_dev, nu_fits=nu_fits, fit_flags=flags, log10_tau=log10_tau,
              per_channel="device", seed_ns=args.seed_ns)

    def step():
        res = eng.fit_batch(data, freqs, P, x0, **kw)
        rec = ppdist.pack_records(res)
        out = ppdist.gather_records(rec, device=device)   # one RCCL gather
        return res, out

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.set_option("profile", 1)
    eng.kernel_times(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, gathered = step()
    fence()
    elapsed = time.perf_counter() - t0
    eng.set_option("profile", 0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ktimes = eng.kernel_times()

    if rank == 0:
        total_fits = nsub * world * args.steps
        value = total_fits / elapsed
        # ---- roofline of the dominant kernel family (HIP events on the
        # engine's own stream, recorded inside the timed region) ----
        fam = max((k for k in ktimes if ktimes[k][1] > 0), key=lambda k: ktimes[k][0])
        fam_s, fam_n = ktimes[fam]
        per_step_s = fam_s / args.steps
        abytes = algorithmic_bytes_per_fit(C, B, s_bytes, nsub)
        achieved = abytes * nsub / per_step_s / 1e9
        # HBM bytes per launch of that kernel from the PMC passes (FETCH_SIZE x2 +
        # WRITE_SIZE, collected separately with rocprofv3 --pmc and committed under
        # profiles/); null when no matching profile exists
        traffic, co_limit = None, None
        try:
            tp = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if (tp["workload"] == args.workload and tp["input_dtype"] == args.input_dtype
                    and tp["kernel"] == fam):
                traffic = tp["hbm_bytes_per_fit"] * nsub
                if "valu_issue_frac_per_wave" in tp:
                    # what actually limits the kernel (counter passes under profiles/)
                    co_limit = {"resource": "f64 VALU issue",
                                "busy_frac": round(tp["valu_issue_frac_per_wave"] *
                                                   tp.get("waves_per_simd", 1), 3),
                                "source": tp.get("counters_source")}
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": fam, "co_limit": co_limit,
                    "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": abytes * nsub,
                    "algorithmic_bytes_per_fit": abytes,
                    "fits_per_launch_group": nsub,
                    "launches_per_step": fam_n / args.steps,
                    "ms_per_step_in_kernel": round(1e3 * per_step_s, 4),
                    "all_kernels_ms_per_step": {
                        k: round(1e3 * v[0] / args.steps, 4) for k, v in ktimes.items()
                        if v[1] > 0}}
        line = {"metric": "subint_fits_per_sec", "value": round(value, 2),
                "unit": "fits/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": args.workload, "note": note,
                           "nsub_per_gpu_per_step": nsub, "nchan": C, "nbin": B,
                           "fit_flags": flags, "input_dtype": args.input_dtype,
                           "bytes_per_sample_resident": s_bytes, "dm0": args.dm0,
                           "sigma": args.sigma, "model_harmonics_kept": nharm,
                           "device_phase_seed_ns": args.seed_ns,
                           "parallelism": "subint shards, %d rank(s), 1 gather" % world},
                "roofline": roofline,
                "convergence": {"nfeval_mean": float(np.mean(res["nfeval"])),
                                "nfeval_max": int(np.max(res["nfeval"])),
                                "return_codes": {str(k): int(v) for k, v in zip(
                                    *np.unique(res["return_code"], return_counts=True))}}}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(data, freqs, P, x0, errs, nu_fit, flags,
                                                log10_tau, res, args.cpu_sample, model)
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(data, freqs, P, x0, errs, nu_fit, flags, log10_tau, res, nsample, model):
    """Time the CPU oracle (a NumPy/SciPy port of the reference algorithm) on a
    bounded sample of the very batch the GPU fitted, and report parity on it."""
    from oracle import pptoas_oracle as orc
    budget_s, cap = 12.0, min(64, data.shape[0])
    want = cap if nsample <= 0 else max(1, min(nsample, data.shape[0]))
    t0 = time.perf_counter()
    outs = []
    for i in range(want):
        host = data[i].cpu().numpy().astype(np.float64)
        outs.append(orc.fit_portrait_full(host, model, x0[i], P[i], freqs,
                                          [nu_fit] * 3, [None] * 3, errs[i], flags,
                                          log10_tau=log10_tau))
        # bounded sample: stop when the next fit would overrun the time budget
        el = time.perf_counter() - t0
        if nsample <= 0 and el + el / (i + 1) > budget_s:
            break
    nsample = len(outs)
    dt = time.perf_counter() - t0
    dphi = max(abs(((o.phi - res["params"][i, 0]) + 0.5) % 1.0 - 0.5)
               for i, o in enumerate(outs))
    dDM = max(abs(o.DM - res["params"][i, 1]) for i, o in enumerate(outs))
    return {"value": round(nsample / dt, 5), "unit": "fits/s", "cores": 1,
            "kind": "port", "host_cpu_count": os.cpu_count(),
            "sample": "%d subint(s) of the timed batch, whole fit_portrait_full "
                      "(oracle/pptoas_oracle.py), %.1f s" % (nsample, dt),
            "parity_on_sample": {"max_abs_dphi": dphi, "max_abs_dDM": dDM}}


if __name__ == "__main__":
    main()
