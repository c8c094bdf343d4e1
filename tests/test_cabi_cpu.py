"""CPU-side checks of the boundary: the C-ABI library loads, exports every
symbol include/pp_toas.h declares, and refuses to work without a GPU instead
of silently falling back."""
import os
import re

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
