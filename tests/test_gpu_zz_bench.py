"""The multi-GPU bench path as far as one GPU allows: `bench.py --total-nsub 2500 --nsub
1024` (configs[4]'s strong-scaling flow: contiguous shard, device-generated sub-batches,
a ragged last one, records kept in HBM, one gather) run in a fresh child process --
started by tests/conftest.py before this process touched the GPU -- and its records
checked against fits made directly in this process.  (Named to run last: the child
works beside the other GPU tests.)"""
import argparse
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_strong_scaling_bench_in_a_child_process():
    import torch
    from tests.conftest import BENCH_CHILD, BENCH_CHILD_ARGS
    import bench
    from pulseportraiture_amd import dist as ppdist
    from pulseportraiture_amd.engine import Engine
    if os.environ.get("PP_NO_BENCH_CHILD"):
        pytest.skip("PP_NO_BENCH_CHILD")
    assert BENCH_CHILD.get("proc") is not None, "conftest did not start the bench child (GPU visible?)"
    rc = BENCH_CHILD["proc"].wait(timeout=800)
    BENCH_CHILD["out"].close(); BENCH_CHILD["err"].close()
    text = open(os.path.join(BENCH_CHILD["tmp"], "line.json")).read().strip()
    assert rc == 0 and text, open(os.path.join(BENCH_CHILD["tmp"], "stderr.txt")).read()[-2000:]
    line = json.loads(text.splitlines()[-1])
    total, nsub = 2500, 1024
    assert line["scaling"] == "strong" and line["n_gpus"] == 1 and line["metric"] == "subint_fits_per_sec"
    cfg = line["config"]
    assert cfg["total_nsub"] == total and cfg["fits_per_rank"] == [total] and cfg["sub_batch"] == nsub
    assert cfg["sub_batches_rank0"] == [1024, 1024, 452]              # (the ragged last sub-batch)
    assert line["gathered_records"]["rows"] == total
    # (the rate is no measurement here: five bench children and the test process share the box's one GPU)
    assert line["value"] > 1e3 and line["max_abs_dDM_over_err"] < 6.0
    assert cfg["resident_sub_batches"] == 2 and cfg["steps_in_flight"] == 3
    rec = np.load(BENCH_CHILD["records"])
    assert rec.shape == (total, ppdist.RECORD_WIDTH)
    np.testing.assert_allclose(rec.sum(axis=0), line["gathered_records"]["column_sums"], rtol=1e-12)
    assert (rec[:, 17] == 2).all() and (rec[:, 16] >= 3).all()
    # the same subints fitted directly here: the first 48 of the job and the ragged tail
    # [2048, 2500), generated from their global indices (bench.Batch, the child's defaults)
    ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05,
                            truth_guesses=False, measured_noise=False, method="trust-ncg")
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    for first, n in ((0, 48), (2048, total - 2048)):
        b = bench.Batch(eng, ns, dev, cfg["workload"], n, "f64", first)
        out = torch.zeros((n, ppdist.RECORD_WIDTH), dtype=torch.float64, device=dev)
        r = b.fit(records=out)
        mine = out.cpu().numpy()
        # (bit for bit: the child fitted these subints in sub-batches of 1024 / 452, this process in batches of
        # 48 / 452 -- a subint's answer does not depend on the batch)
        np.testing.assert_array_equal(mine, rec[first:first + n])
        np.testing.assert_array_equal(r["params"], rec[first:first + n, :5])
        b.free()
    eng.close()


@pytest.mark.timeout(900)
def test_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` -- the plain command, NO launcher in the test: bench.py starts
    torch.distributed.run itself as a child process (bench.self_launch) and hands on the one JSON
    line --, the two ranks sharing this box's GPU and
    talking over gloo (PP_BENCH_SHARE_GPU=1; RCCL wants one device per rank): the N > 1 path of
    both modes with the real engine.  Strong scaling (600 subints of configs[1] in contiguous
    shards of 300, sub-batches 256 + 44, ONE gather): every gathered record against fits made
    directly in this process.  Weak scaling (128 subints per rank and step): the line's
    bookkeeping."""
    import torch
    from tests.conftest import BENCH_CHILD
    import bench
    from pulseportraiture_amd import dist as ppdist
    from pulseportraiture_amd.engine import Engine
    if os.environ.get("PP_NO_BENCH_CHILD"):
        pytest.skip("PP_NO_BENCH_CHILD")
    lines = {}
    for tag in ("strong", "weak"):
        ch = BENCH_CHILD.get("two_" + tag)
        assert ch is not None, "conftest did not start the two-rank bench (GPU visible?)"
        rc = ch["proc"].wait(timeout=800)
        ch["out"].close(); ch["err"].close()
        text = open(os.path.join(BENCH_CHILD["tmp"], "line2_%s.json" % tag)).read().strip()
        err = open(os.path.join(BENCH_CHILD["tmp"], "stderr2_%s.txt" % tag)).read()
        assert rc == 0 and text, err[-3000:]
        out_lines = [ln for ln in text.splitlines() if ln.strip()]
        assert len(out_lines) == 1, out_lines[:5]            # exactly the one JSON line on stdout
        assert "starting" in err and "torch.distributed.run" in err      # (bench.py launched its own ranks)
        lines[tag] = json.loads(out_lines[0])
    line = lines["strong"]
    total = 600
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    cfg = line["config"]
    assert cfg["fits_per_rank"] == [300, 300] and cfg["sub_batches_rank0"] == [256, 44]
    assert line["gathered_records"]["rows"] == total and line["value"] > 0
    assert line["wall_s"] >= 1e-3 * line["ms_per_step"] and 0 < line["fits_per_s_all_in"] <= line["value"]
    rec = np.load(BENCH_CHILD["two_strong"]["records"])
    assert rec.shape == (total, ppdist.RECORD_WIDTH) and (rec[:, 17] == 2).all()
    ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05,
                            truth_guesses=False, measured_noise=False, method="trust-ncg")
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    b = bench.Batch(eng, ns, dev, cfg["workload"], total, "f64", 0)       # (subints by their global indices)
    out = torch.zeros((total, ppdist.RECORD_WIDTH), dtype=torch.float64, device=dev)
    b.fit(records=out)
    mine = out.cpu().numpy()
    np.testing.assert_array_equal(mine, rec)            # rank 0's and rank 1's, bit for bit (sub-batches 256 + 44 there, 600 here)
    b.free(); eng.close()
    w = lines["weak"]
    assert w["scaling"] == "weak" and w["n_gpus"] == 2 and w["steps"] == 2
    assert w["config"]["nsub_per_gpu_per_step"] == 128
    assert abs(w["value"] - 2 * 128 * 2 / (w["ms_per_step"] * 2e-3)) < 5e-3 * w["value"]     # all ranks' fits / max time (ms rounded)
    assert w["gathered_records"]["rows"] == 2 * 2 * 128


@pytest.mark.timeout(900)
def test_rccl_backend_with_one_rank():
    """RCCL as far as one GPU allows: bench.py under torch.distributed.run with ONE rank and the
    "nccl" backend (= RCCL on ROCm) -- process-group creation bound to the device, the barriers
    around the timed region, the max-over-ranks all_reduce and the gather of DEVICE record tensors
    all go through RCCL, and its version banner must not reach stdout, which carries exactly one
    JSON line.  Weak mode (256 subints of the headline shape, 2 steps) and configs[4]'s strong
    mode (1500 subints in sub-batches of 256, ragged tail, one gather): rows, bookkeeping, and
    the records against fits made directly in this process."""
    import torch
    from tests.conftest import BENCH_CHILD
    import bench
    from pulseportraiture_amd import dist as ppdist
    from pulseportraiture_amd.engine import Engine
    if os.environ.get("PP_NO_BENCH_CHILD"):
        pytest.skip("PP_NO_BENCH_CHILD")
    lines = {}
    for tag in ("weak", "strong"):
        ch = BENCH_CHILD.get("rccl_" + tag)
        assert ch is not None, "conftest did not start the one-rank RCCL bench (GPU visible?)"
        rc = ch["proc"].wait(timeout=800)
        ch["out"].close(); ch["err"].close()
        text = open(os.path.join(BENCH_CHILD["tmp"], "line1_%s.json" % tag)).read().strip()
        err = open(os.path.join(BENCH_CHILD["tmp"], "stderr1_%s.txt" % tag)).read()
        assert rc == 0 and text, err[-3000:]
        out_lines = [ln for ln in text.splitlines() if ln.strip()]
        assert len(out_lines) == 1, out_lines[:5]            # (no RCCL banner, nothing but the line)
        lines[tag] = json.loads(out_lines[0])
    w = lines["weak"]
    assert w["scaling"] == "weak" and w["n_gpus"] == 1 and w["steps"] == 2
    assert w["config"]["nsub_per_gpu_per_step"] == 256 and "1 rank(s)" in w["config"]["parallelism"]
    assert w["gathered_records"]["rows"] == 2 * 256 and w["value"] > 1e3
    s = lines["strong"]
    total = 1500
    assert s["scaling"] == "strong" and s["n_gpus"] == 1
    assert s["config"]["fits_per_rank"] == [total] and s["config"]["sub_batches_rank0"] == [256] * 5 + [220]
    assert s["gathered_records"]["rows"] == total
    rec = np.load(BENCH_CHILD["rccl_strong"]["records"])
    assert rec.shape == (total, ppdist.RECORD_WIDTH) and (rec[:, 17] == 2).all()
    np.testing.assert_allclose(rec.sum(axis=0), s["gathered_records"]["column_sums"], rtol=1e-12)
    ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05,
                            truth_guesses=False, measured_noise=False, method="trust-ncg")
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    # the weak job's records are those of subints [0, 256) (both steps fit the same resident batch):
    # its checksum against a direct fit; the strong job's ragged tail [1280, 1500) record by record
    b = bench.Batch(eng, ns, dev, w["config"]["workload"], 256, "f64", 0)
    out = torch.zeros((256, ppdist.RECORD_WIDTH), dtype=torch.float64, device=dev)
    b.fit(records=out)
    mine = out.cpu().numpy()
    np.testing.assert_allclose(2.0 * mine[:, :3].sum(axis=0), w["gathered_records"]["checksum"], rtol=1e-12)
    np.testing.assert_allclose(mine[:, :13], rec[:256, :13], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(mine[:, 13:16], rec[:256, 13:16], rtol=1e-11)
    b.free()
    b = bench.Batch(eng, ns, dev, s["config"]["workload"], 220, "f64", 1280)
    out = torch.zeros((220, ppdist.RECORD_WIDTH), dtype=torch.float64, device=dev)
    b.fit(records=out)
    mine = out.cpu().numpy()
    # (the child formed its guesses in batches of 256, this process in one of 220: since round 5 the channel
    # runs of pp_reference_phase_seed depend on the band alone, so the guesses -- and with them every output --
    # are the same bits whatever the batch; until round 4 up to 4 of the 220 rows moved by ~1e-9 rot)
    tail = rec[1280:]
    moved = (np.abs(mine[:, :13] - tail[:, :13]) > 1e-13 * np.abs(tail[:, :13]) + 1e-15).any(axis=1)
    assert moved.sum() == 0, int(moved.sum())
    np.testing.assert_array_equal(mine[:, 16:], tail[:, 16:])
    np.testing.assert_allclose(mine[:, 13:16], tail[:, 13:16], rtol=1e-11)
    b.free(); eng.close()


@pytest.mark.timeout(900)
def test_rccl_backend_with_two_ranks():
    """N = 2 for real -- one rank per GPU over the "nccl" backend (RCCL over xGMI) -- on a node that has two
    GPUs; the one-GPU boxes of this pool skip it.  Weak mode (256 subints per rank and step): rows and
    bookkeeping.  configs[4]'s strong mode (1500 subints in shards of 750, sub-batches 256 + 256 + 238):
    every gathered record against fits made directly in this process -- bit for bit, since a subint's
    answer does not depend on how shards and sub-batches fall (tests/test_gpu_y_batch_independence.py)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    from tests.conftest import BENCH_CHILD
    import bench
    from pulseportraiture_amd import dist as ppdist
    from pulseportraiture_amd.engine import Engine
    if os.environ.get("PP_NO_BENCH_CHILD"):
        pytest.skip("PP_NO_BENCH_CHILD")
    lines = {}
    for tag in ("weak", "strong"):
        ch = BENCH_CHILD.get("rccl2_" + tag)
        assert ch is not None, "conftest did not start the two-rank RCCL bench"
        rc = ch["proc"].wait(timeout=800)
        ch["out"].close(); ch["err"].close()
        text = open(os.path.join(BENCH_CHILD["tmp"], "lineN2_%s.json" % tag)).read().strip()
        err = open(os.path.join(BENCH_CHILD["tmp"], "stderrN2_%s.txt" % tag)).read()
        assert rc == 0 and text, err[-3000:]
        out_lines = [ln for ln in text.splitlines() if ln.strip()]
        assert len(out_lines) == 1, out_lines[:5]
        lines[tag] = json.loads(out_lines[0])
    w = lines["weak"]
    assert w["scaling"] == "weak" and w["n_gpus"] == 2 and w["steps"] == 2
    assert w["gathered_records"]["rows"] == 2 * 2 * 256 and "2 rank(s)" in w["config"]["parallelism"]
    assert abs(w["value"] - 2 * 256 * 2 / (w["ms_per_step"] * 2e-3)) < 5e-3 * w["value"]
    s = lines["strong"]
    total = 1500
    assert s["scaling"] == "strong" and s["n_gpus"] == 2
    assert s["config"]["fits_per_rank"] == [750, 750] and s["config"]["sub_batches_rank0"] == [256, 256, 238]
    rec = np.load(BENCH_CHILD["rccl2_strong"]["records"])
    assert rec.shape == (total, ppdist.RECORD_WIDTH) and (rec[:, 17] == 2).all()
    np.testing.assert_allclose(rec.sum(axis=0), s["gathered_records"]["column_sums"], rtol=1e-12)
    ns = argparse.Namespace(seed=20260101, dm0=34.56789, dm_offset=[3e-4, 2e-4], sigma=0.05,
                            truth_guesses=False, measured_noise=False, method="trust-ncg")
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    for first, n in ((0, 300), (750, 256), (1262, 238)):       # rank 0's start, rank 1's first and last sub-batch
        b = bench.Batch(eng, ns, dev, s["config"]["workload"], n, "f64", first)
        out = torch.zeros((n, ppdist.RECORD_WIDTH), dtype=torch.float64, device=dev)
        b.fit(records=out)
        np.testing.assert_array_equal(out.cpu().numpy(), rec[first:first + n])
        b.free()
    eng.close()
