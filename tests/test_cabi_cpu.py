"""CPU-side checks of the boundary: the C-ABI library loads, exports every
symbol include/pp_toas.h declares, and refuses to work without a GPU instead
of silently falling back."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from pulseportraiture_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "pp_toas.h")).read()
    declared = set(re.findall(r"\b(pp_[a-z_0-9]+)\s*\(", header))
    declared -= {"pp_ctx"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export %s" % name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert lib.pp_abi_version() == _lib.ABI_VERSION


def test_no_silent_cpu_fallback():
    """Without a GPU pp_create must fail with an error string; with one it
    must succeed.  Either way nothing routes through the oracle."""
    import ctypes as C
    from pulseportraiture_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.pp_create(0, C.byref(ctx))
    if rc == 0:
        assert lib.pp_destroy(ctx) == 0
    else:
        assert rc < 0 and len(_lib.last_error()) > 0


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pulseportraiture_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), \
                    "%s mentions the oracle" % f


def test_missing_library_fails_loudly(monkeypatch):
    from pulseportraiture_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpptoas_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_no_scratch_traffic_inside_the_row_loops_of_the_transforms():
    """The metadata's "spilled VGPRs" of the 2048-bin transform kernels cannot tell the saves and restores around
    their one call to tail_work (executed once per ticket a wave draws) from spills inside the row loop (executed per
    row; a scratch load queues behind the prefetched row like every other vector-memory access -- worth 7 % once).  The
    disassembly can (tools/kernel_resources.py --loops): a loop is a backward branch, and no scratch instruction of any
    k_xspec_q* instantiation may sit in a loop that contains no call.  The kernels that carry tickets must still have
    their call; the others must have no scratch at all."""
    import shutil
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or shutil.which("objcopy") is None:
        pytest.skip("no llvm-objdump / objcopy")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    res = kernel_resources.loop_scratch(pat="k_xspec_q")
    assert len(res) >= 18, sorted(res)
    carriers = [k for k in res if k.startswith(("k_xspec_q1024<", "k_xspec_qf<1024,")) or
                (k.startswith("k_xspec_qr1024<") and k.endswith("false>"))]
    assert len(carriers) == 10, carriers
    for k, v in res.items():
        assert v["scratch_in_loops"] == 0, (k, v)
        if k in carriers:
            assert v["calls"] == 2 and v["scratch_total"] > 0, (k, v)       # (the ticket between two rows, the drain at the end)
        else:
            assert v["calls"] == 0 and v["scratch_total"] == 0, (k, v)
