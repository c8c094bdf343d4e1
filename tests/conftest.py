import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the strong-scaling bench run the GPU suite checks (tests/test_gpu_zz_bench.py): started
# here, as a CHILD process, before this process has touched the GPU (a process that has
# initialised the GPU must not start other programs on this pool) -- it runs beside the
# first tests and is collected by the last one
BENCH_CHILD = {}
# (--group 2: two resident sub-batches of 68.7 GB -- the children and the tests share one GPU)
BENCH_CHILD_ARGS = ["--total-nsub", "2500", "--nsub", "1024", "--group", "2", "--no-cpu-baseline"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_finish(session):
    # only when a test that collects a child is going to run
    want = ("test_strong_scaling_bench_in_a_child_process", "test_two_ranks_sharing_the_gpu",
            "test_rccl_backend_with_one_rank", "test_rccl_backend_with_two_ranks")
    if not any(w in it.nodeid for it in session.items for w in want) or os.environ.get("PP_NO_BENCH_CHILD"):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:        # (counting devices does not initialise the GPU)
            return
    except Exception:
        return
    tmp = tempfile.mkdtemp(prefix="pp_bench_child_")
    BENCH_CHILD.update(tmp=tmp)
    if any(want[0] in it.nodeid for it in session.items):
        out, err = open(os.path.join(tmp, "line.json"), "w"), open(os.path.join(tmp, "stderr.txt"), "w")
        recs = os.path.join(tmp, "records.npy")
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_CHILD_ARGS +
                                ["--dump-records", recs], stdout=out, stderr=err, cwd=ROOT)
        BENCH_CHILD.update(proc=proc, records=recs, out=out, err=err)


    # ... and the N = 2 path with the real engine: `python bench.py --gpus 2`, two ranks that share
    # this box's GPU and talk over gloo (PP_BENCH_SHARE_GPU=1), strong and weak scaling
    if any("test_two_ranks_sharing_the_gpu" in it.nodeid for it in session.items):
        import socket
        for tag, extra in (("strong", ["--total-nsub", "600", "--nsub", "256"]), ("weak", ["--nsub", "128", "--steps", "2"])):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            o = open(os.path.join(tmp, "line2_%s.json" % tag), "w")
            e = open(os.path.join(tmp, "stderr2_%s.txt" % tag), "w")
            r2 = os.path.join(tmp, "records2_%s.npy" % tag)
            # the PLAIN command, no launcher in front of it: bench.py starts torch.distributed.run itself as a
            # child process (bench.self_launch) -- the form the driver uses for its N = 1 line
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"),
                   "--gpus", "2", "--workload", "cfg2-512x1024-phiDM", "--no-cpu-baseline"] + extra
            if tag == "strong":
                cmd += ["--dump-records", r2]
            env2 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
            p2 = subprocess.Popen(cmd, stdout=o, stderr=e, cwd=ROOT, env=dict(env2, PP_BENCH_SHARE_GPU="1"))
            BENCH_CHILD["two_" + tag] = dict(proc=p2, out=o, err=e, records=r2)


    # ... and RCCL itself as far as one GPU allows: ONE rank under torch.distributed.run with the "nccl"
    # backend (communicator creation with its banner kept off stdout, barriers, the gather of device
    # tensors), weak and strong modes
    if any("test_rccl_backend_with_one_rank" in it.nodeid for it in session.items):
        import socket
        for tag, extra in (("weak", ["--nsub", "256", "--steps", "2", "--no-other-workloads"]),
                           ("strong", ["--total-nsub", "1500", "--nsub", "256"])):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            o = open(os.path.join(tmp, "line1_%s.json" % tag), "w")
            e = open(os.path.join(tmp, "stderr1_%s.txt" % tag), "w")
            r1 = os.path.join(tmp, "records1_%s.npy" % tag)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                   "--gpus", "1", "--no-cpu-baseline"] + extra
            if tag == "strong":
                cmd += ["--dump-records", r1]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.pop("PP_BENCH_SHARE_GPU", None)
            p1 = subprocess.Popen(cmd, stdout=o, stderr=e, cwd=ROOT, env=env)
            BENCH_CHILD["rccl_" + tag] = dict(proc=p1, out=o, err=e, records=r1)


    # ... and, on a node with two GPUs or more, the real thing: TWO ranks, one GPU each, over the "nccl" backend
    # (RCCL over xGMI).  The one-GPU boxes of this pool skip it (tests/test_gpu_zz_bench.py).
    if any("test_rccl_backend_with_two_ranks" in it.nodeid for it in session.items) and torch.cuda.device_count() >= 2:
        import socket
        for tag, extra in (("weak", ["--nsub", "256", "--steps", "2", "--no-other-workloads"]),
                           ("strong", ["--total-nsub", "1500", "--nsub", "256"])):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            o = open(os.path.join(tmp, "lineN2_%s.json" % tag), "w")
            e = open(os.path.join(tmp, "stderrN2_%s.txt" % tag), "w")
            r1 = os.path.join(tmp, "recordsN2_%s.npy" % tag)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                   "--gpus", "2", "--no-cpu-baseline"] + extra
            if tag == "strong":
                cmd += ["--dump-records", r1]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.pop("PP_BENCH_SHARE_GPU", None)
            p1 = subprocess.Popen(cmd, stdout=o, stderr=e, cwd=ROOT, env=env)
            BENCH_CHILD["rccl2_" + tag] = dict(proc=p1, out=o, err=e, records=r1)


def pytest_sessionfinish(session, exitstatus):
    procs = [BENCH_CHILD.get("proc")] + [v.get("proc") for k, v in BENCH_CHILD.items()
                                         if k.startswith("two_") or k.startswith("rccl_") or k.startswith("rccl2_")]
    for proc in procs:
        if proc is not None and proc.poll() is None:
            proc.kill()
            proc.wait()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
