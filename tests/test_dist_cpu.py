"""N>1 path on CPU: contiguous subint shards and the single gather of result
records, world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest

from pulseportraiture_amd import dist as ppdist


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 100000, 12501):
        for world in (1, 2, 3, 8):
            spans = [ppdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_result(lo, hi):
    n = hi - lo
    idx = np.arange(lo, hi, dtype=np.float64)
    return dict(params=np.outer(idx, [1, 2, 3, 4, 5.0]),
                param_errs=np.outer(idx, [0.1, 0.2, 0.3, 0.4, 0.5]),
                nu_refs=np.outer(idx, [10, 20, 30.0]), chi2=idx * 7, red_chi2=idx / 3,
                snr=idx + 0.5, nfeval=np.full(n, 4, dtype=np.int32),
                return_code=np.full(n, 2, dtype=np.int32))


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = [b - a for a, b in (ppdist.shard_range(n, r, world) for r in range(world))]
        lo, hi = ppdist.shard_range(n, rank, world)
        rec = ppdist.pack_records(_fake_result(lo, hi))
        out = ppdist.gather_records(rec, counts=counts)
        if rank == 0:
            q.put(out)
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_records_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n, world = 11, 2            # ragged: 6 + 5
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = ppdist.pack_records(_fake_result(0, n))
    np.testing.assert_array_equal(out, want)
    f = ppdist.unpack_records(out)
    np.testing.assert_array_equal(f["DM"], np.arange(n) * 2.0)
    assert set(f) == set(ppdist.RECORD_FIELDS)


def _worker_tensor(rank, world, port, n, q):
    """The bench's N > 1 data path with the shapes Engine.fit_batch(records=...)
    produces: every rank keeps [steps][nsub][RECORD_WIDTH] float64 tensors and ONE
    gather at the end brings them to rank 0 (torch tensors in, tensor out)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = [b - a for a, b in (ppdist.shard_range(n, r, world) for r in range(world))]
        lo, hi = ppdist.shard_range(n, rank, world)
        rec = torch.from_numpy(ppdist.pack_records(_fake_result(lo, hi)))
        assert rec.dtype == torch.float64 and rec.shape == (hi - lo, ppdist.RECORD_WIDTH)
        out = ppdist.gather_records(rec, counts=counts)
        if rank == 0:
            assert torch.is_tensor(out)
            q.put((out.numpy(), ppdist.records_checksum(out)))
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_tensor_records_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n, world = 13, 2
    procs = [ctx.Process(target=_worker_tensor, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, cs = q.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = ppdist.pack_records(_fake_result(0, n))
    np.testing.assert_array_equal(out, want)
    assert cs["rows"] == n
    np.testing.assert_allclose(cs["column_sums"], want.sum(axis=0))


class _StubBatch(object):
    """What bench.strong_scaling needs of a resident batch, without a GPU: the 'fit' of
    subint g (global index) is a fixed function of g, written into the records tensor
    the way pp_fit_out.records_dev is."""
    guess = "stub"

    def __init__(self, workload, nsub, first):
        self.nsub, self.calls = nsub, []
        self.generate(first)

    def generate(self, first):
        self.first = first
        self.inj = np.zeros((self.nsub, 3))
        self.inj[:, 1] = 30.0 + 1e-3 * (first + np.arange(self.nsub))

    def fit(self, records=None, n=None):
        import torch
        n = self.nsub if n is None else n
        res = _fake_result(self.first, self.first + n)
        res["params"][:, 1] = self.inj[:n, 1] + 0.5 * res["param_errs"][:, 1]
        if records is not None:
            assert tuple(records.shape) == (n, ppdist.RECORD_WIDTH)
            records.copy_(torch.from_numpy(ppdist.pack_records(res)))
            self.calls.append((self.first, n))
        return res


def _worker_strong(rank, world, port, total, nsub, q):
    import argparse
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args = argparse.Namespace(workload="cfg2-512x1024-phiDM", nsub=nsub, total_nsub=total, input_dtype="f64",
                                  method="trust-ncg", dump_records=None)
        made = []

        def make_batch(workload, n, first):
            made.append(_StubBatch(workload, n, first))
            return made[-1]
        line = bench.strong_scaling(args, make_batch, lambda: None, dist.barrier, "cpu", rank, world, True)
        lo, hi = ppdist.shard_range(total, rank, world)
        # this rank fitted exactly its contiguous shard, in sub-batches of nsub, the last one ragged
        want, g = [], lo
        while g < hi:
            want.append((g, min(nsub, hi - g)))
            g += nsub
        # (the shard is worked through in groups of up to three resident sub-batches: the calls of all of them)
        assert len(made) <= 3
        calls = sorted(c for m in made for c in m.calls)
        assert calls == want, (calls, want)
        q.put((rank, line))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_strong_scaling_shards_and_gathers_world2_gloo():
    """bench.strong_scaling (configs[4]'s flow: contiguous shards, sub-batches, ragged
    tails, ONE gather, max-over-ranks timing) driven on CPU with a stub batch, world 2."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, nsub, world = 2501, 600, 2           # shards 1251 + 1250; sub-batches 600, 600, 51 / 600, 600, 50
    procs = [ctx.Process(target=_worker_strong, args=(r, world, port, total, nsub, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is None
    line = got[0]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["fits_per_rank"] == [1251, 1250]
    assert line["config"]["sub_batches_rank0"] == [600, 600, 51]
    assert line["gathered_records"]["rows"] == total
    want = _fake_result(0, total)
    want["params"][:, 1] = 30.0 + 1e-3 * np.arange(total) + 0.5 * want["param_errs"][:, 1]
    np.testing.assert_allclose(line["gathered_records"]["column_sums"], ppdist.pack_records(want).sum(axis=0),
                               rtol=1e-13)
    assert abs(line["max_abs_dDM_over_err"] - 0.5) < 1e-9
    assert line["value"] > 0 and line["ms_per_step"] > 0
    # the all-in wall time (generation, warm-up and ramp fits, fits, gather) beside the timed figure
    assert line["wall_s"] >= 1e-3 * line["ms_per_step"] and 0 < line["fits_per_s_all_in"] <= line["value"]


@pytest.mark.timeout(180)
def test_plain_bench_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher around it (the form the driver uses at N = 1) must start
    the N ranks itself -- torch.distributed.run as a CHILD process -- and hand on exactly the one JSON line
    and the return code.  Here with a stand-in for the ranks' program (no GPU): two ranks see RANK /
    WORLD_SIZE / MASTER_ADDR, rank 0 prints a banner and the line, and a failing rank's code comes back."""
    import json
    import subprocess
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and '--gpus' in sys.argv\n"
        "if '--fail' in sys.argv and os.environ['RANK'] == '1':\n"
        "    sys.exit(7)\n"
        "if os.environ['RANK'] == '0':\n"
        "    print('RCCL version banner'); print(json.dumps({'metric': 'm', 'n_gpus': 2, 'argv': sys.argv[1:]}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(2, sys.argv[1:], script=%r))" % (root, str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code, "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       env=env, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["argv"] == ["--gpus", "2", "--steps", "2"]
    assert "RCCL version banner" in r.stderr            # (kept off stdout)
    r = subprocess.run([sys.executable, "-c", code, "--gpus", "2", "--fail"], capture_output=True, text=True,
                       env=env, timeout=150)
    assert r.returncode != 0


def test_bench_main_takes_the_self_launch_branch_before_torch():
    """bench.main() must take that branch before anything imports torch or touches the GPU."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    body = src[src.index("def main():"):]
    assert 0 < body.index("self_launch(args.gpus") < body.index("import torch")
    assert "start_cpu_pool()" in body and body.index("self_launch(args.gpus") < body.index("start_cpu_pool()")


def test_record_layout_matches_the_c_abi():
    """RECORD_FIELDS is the layout k_finalize writes into pp_fit_out.records_dev
    (include/pp_toas.h PP_RECORD_WIDTH)."""
    import re
    from pulseportraiture_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                            "pp_toas.h")).read()
    assert int(re.search(r"#define PP_RECORD_WIDTH (\d+)", hdr).group(1)) == ppdist.RECORD_WIDTH
    assert _lib.PP_RECORD_WIDTH == ppdist.RECORD_WIDTH == 18
