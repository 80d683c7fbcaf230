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
    assert lib.hm_abi_version() == 2


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


def test_host_model_mirrors_the_surface_the_notebook_uses():
    """The simulator members notebooks/HistoryMatch.py touches (found by running it on the oracle: oracle/run_reference_script.py)
    exist on the GPU-backed model and its grid helpers agree with the oracle's -- no device is needed for any of them."""
    import numpy as np

    from historymatching_amd.ressim import ResSim
    from oracle.ressim import ResSim as OracleResSim

    gm, om = ResSim(20, 12, 2, 1), OracleResSim(20, 12, 2, 1)
    for m in (gm, om):
        m.inj_xy = [[1.0, 0.5]]
        m.prd_xy = [[0.24, 0.12], [1.74, 0.87]]
        m.inj_rates, m.prd_rates = [[1]], np.ones((2, 1)) / 2
    surface = ["Nx", "Ny", "Lx", "Ly", "Nxy", "shape", "mesh", "prd_xy", "inj_xy", "inj_rates", "prd_rates", "nInj", "nPrd", "sim",
               "xy2ind", "xy2sub", "sub2ind", "sub2xy", "ind2sub", "ind2xy", "K", "domain"]
    for name in surface:
        assert hasattr(gm, name), name
    assert (gm.Nxy, gm.shape, gm.nInj, gm.nPrd) == (om.Nxy, om.shape, om.nInj, om.nPrd)
    assert np.array_equal(gm.prd_xy, om.prd_xy) and np.array_equal(gm.inj_xy, om.inj_xy)  # collocated to cell centres
    for a, b in zip(gm.mesh, om.mesh):
        assert np.array_equal(a, b)
    rng = np.random.RandomState(0)
    x, y = rng.rand(50) * 2, rng.rand(50)
    assert np.array_equal(gm.xy2ind(x, y), om.xy2ind(x, y))
    ind = rng.randint(0, gm.Nxy, 30)
    assert np.array_equal(np.asarray(gm.ind2xy(ind)), np.asarray(om.ind2xy(ind)))
    assert np.array_equal(np.asarray(gm.ind2sub(ind)), np.asarray(om.ind2sub(ind)))
    ix, iy = gm.ind2sub(ind)
    assert np.array_equal(gm.sub2ind(ix, iy), ind) and np.array_equal(np.asarray(gm.sub2xy(ix, iy)), np.asarray(om.sub2xy(ix, iy)))
    # every task of the reference deep-copies the model (HistoryMatch.py:360) and its process pool pickles it (utils.py:211)
    import copy
    import pickle

    gm.K = 0.1 + np.exp(rng.randn(gm.Nxy))
    for clone in (copy.deepcopy(gm), pickle.loads(pickle.dumps(gm))):
        assert np.array_equal(clone.K, gm.K) and np.array_equal(clone.prd_xy, gm.prd_xy) and clone.nPrd == gm.nPrd
        clone.K = clone.K * 2
        assert not np.array_equal(clone.K, gm.K)  # isolated state


def test_member_block_rule_is_host_logic():
    """forward.default_blocks: how many member blocks (HIP streams) a device-resident ensemble is split into -- three from 768 members, two
    from 512 on the 128 x 128 kernels, one everywhere else (profiles/r05/blocks_time.txt: the larger grids gain nothing)."""
    from historymatching_amd.forward import default_blocks
    from historymatching_amd.ressim import ResSim

    m128, m256, m20 = ResSim(128, 128, 2, 1), ResSim(256, 256, 2, 1), ResSim(20, 20, 2, 1)
    assert [default_blocks(m128, n) for n in (1, 511, 512, 767, 768, 1000, 4096)] == [1, 1, 2, 2, 3, 3, 3]
    assert default_blocks(m256, 4096) == 1 and default_blocks(m20, 1000) == 1 and default_blocks(ResSim(128, 256, 2, 1), 1000) == 1
