"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/s2k.h declares
(no compute without a GPU); the generated stage header is in sync with plan/opdefs.py."""
import ctypes
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g

    g.build()
    return g


def test_header_declares_only_exported_symbols(built):
    hdr = (ROOT / "include" / "s2k.h").read_text()
    names = re.findall(r"\b(s2k_[a-z_0-9]+)\s*\(", hdr)
    assert len(set(names)) >= 8
    lib = ctypes.CDLL(str(built.LIB))
    for n in set(names):
        assert hasattr(lib, n), f"{n} declared in include/s2k.h but not exported by libs2k.so"


def test_ops_header_in_sync():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_opdefs.py"), "--check"])
    assert r.returncode == 0, "include/s2k_ops.h is stale: run tools/gen_opdefs.py"


def test_binding_self_checks_and_error_path(built):
    import s2lc_amd  # noqa: F401
    from s2lc_amd import _lib
    from s2lc_amd.plan import opdefs as D
    from s2lc_amd.plan.program import OP_DTYPE, Program, TRef

    L = _lib.lib()
    assert L.s2k_abi_version() == 2 and L.s2k_op_size() == 256 == OP_DTYPE.itemsize
    assert L.s2k_kind_name(D.KIND["CONV"]).decode() == "CONV" and L.s2k_kind_name(999) is None
    # malformed records are rejected on the host before any launch (safe without a GPU)
    p = Program()
    p.add("AXPY", X=None, Y=None, COUNT=0)
    with pytest.raises(_lib.S2kError):
        _lib.run(p.pack(), _lib.Bases(), 0)
    bad = np.zeros(1, dtype=OP_DTYPE)
    bad["kind"] = 77
    with pytest.raises(_lib.S2kError, match="unknown stage kind"):
        _lib.run(bad, _lib.Bases(), 0)
    p = Program()
    p.add("BN_FINALIZE", STATS=None, GAMMA=TRef(D.BASE["PARAMS"], 0, (4,)), BETA=TRef(D.BASE["PARAMS"], 16, (4,)),
          RM=TRef(D.BASE["BUFS"], 0, (4,)), RV=TRef(D.BASE["BUFS"], 16, (4,)), BNV=TRef(D.BASE["WS"], 0, (16,)),
          COUNT=10, C=4, TRAIN=0, EPS=1e-3, MOM=0.1)
    with pytest.raises(_lib.S2kError, match="null base"):
        _lib.run(p.pack(), _lib.Bases(), 0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import s2lc_amd  # noqa: F401
    from s2lc_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("S2K_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.S2kError, match="no CPU fallback"):
        _lib.lib()
