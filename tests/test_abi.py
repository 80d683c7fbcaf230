"""The C-ABI shared library loads and exports every symbol include/hm_abi.h declares (no GPU needed)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    if not (ROOT / "historymatching_amd" / "libhm_amd.so").exists():
        g.build()
    from historymatching_amd import _lib

    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from historymatching_amd import _lib

    header = (ROOT / "include" / "hm_abi.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hm_abi_version() == 1


def test_no_cpu_fallback_without_device(lib):
    """On a box without a GPU the product path must fail loudly, not compute on the host."""
    import ctypes as C

    from historymatching_amd import _lib

    h = C.c_void_p()
    rc = lib.hm_create(0, C.byref(h))
    if rc == 0:  # running on the GPU box
        lib.hm_destroy(h)
        pytest.skip("GPU present")
    assert b"hip" in lib.hm_last_error().lower() or b"device" in lib.hm_last_error().lower()
    from historymatching_amd.ressim import ResSim
    import numpy as np

    m = ResSim(4, 4)
    m.inj_xy, m.prd_xy, m.inj_rates, m.prd_rates = [[0.1, 0.1]], [[0.9, 0.9]], [[1]], [[1]]
    with pytest.raises(_lib.HmError):
        m.sim(0.1, 1, np.zeros(16))


def test_product_code_never_imports_the_oracle():
    for f in (ROOT / "historymatching_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f
