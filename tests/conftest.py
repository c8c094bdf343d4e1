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
BENCH_CHILD_ARGS = ["--total-nsub", "2500", "--nsub", "1024", "--no-cpu-baseline"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_finish(session):
    # only when the test that collects the child is going to run
    if not any("test_strong_scaling_bench_in_a_child_process" in it.nodeid for it in session.items):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:        # (counting devices does not initialise the GPU)
            return
    except Exception:
        return
    tmp = tempfile.mkdtemp(prefix="pp_bench_child_")
    out, err = open(os.path.join(tmp, "line.json"), "w"), open(os.path.join(tmp, "stderr.txt"), "w")
    recs = os.path.join(tmp, "records.npy")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_CHILD_ARGS +
                            ["--dump-records", recs], stdout=out, stderr=err, cwd=ROOT)
    BENCH_CHILD.update(proc=proc, tmp=tmp, records=recs, out=out, err=err)


def pytest_sessionfinish(session, exitstatus):
    proc = BENCH_CHILD.get("proc")
    if proc is not None and proc.poll() is None:
        proc.kill()
        proc.wait()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
