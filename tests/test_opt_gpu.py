"""Batched NPV objective (historymatching_amd/opt.py: one device batch with a different well configuration per member)
against the per-member CPU restatement (oracle/opt.py) -- SURVEY.md 8f rank 4, Optimise.py:112-125, 170-200."""
import numpy as np
import pytest

from tests.helpers import perms

pytestmark = pytest.mark.gpu

DT, NT = 0.025, 40


def _models(n, rate0=1.5):
    from historymatching_amd.ressim import ResSim as GpuResSim
    from oracle.ressim import ResSim as OracleResSim

    K = 0.1 + np.exp(5 * perms(n, n, 1, seed=23)[0])
    out = []
    for cls in (OracleResSim, GpuResSim):
        m = cls(n, n, 2, 1)
        m.K = K
        near01 = np.array([0.12, 0.87])
        m.inj_xy = [[m.Lx / 2, m.Ly / 2]]
        m.prd_xy = [[x, y] for y in m.Ly * near01 for x in m.Lx * near01]   # Optimise.py:74-90
        m.inj_rates = rate0 * np.ones((1, 1)) / 1
        m.prd_rates = rate0 * np.ones((4, 1)) / 4
        out.append(m)
    return out


@pytest.mark.parametrize("n", [20, 32])
def test_npv_batch_matches_per_member_oracle(n):
    from historymatching_amd.opt import NpvBatch
    from oracle.opt import npv as oracle_npv

    om, gm = _models(n)
    ramp = 1.5 * np.linspace(0.5, 1.5, NT)[None, :]
    params = [
        {},                                                                    # the base configuration
        {"inj_xy": [[0.3, 0.7]]},                                              # Optimise.py:441: injector anywhere on the mesh
        {"inj_xy": [[1.9, 0.05]]},
        {"prd_rates": 1.5 * np.array([[0.4], [0.3], [0.2], [0.1]])},           # Optimise.py:655: rate grid (balanced)
        {"inj_rates": ramp, "prd_rates": np.repeat(ramp / 4, 4, axis=0)},      # time-varying controls (Optimise.py:736-767)
        {"inj_xy": [[2.5, 0.5]]},                                              # outside the domain  -> 0
        {"prd_rates": np.ones((4, 1))},                                        # unbalanced rates     -> 0
        {"prd_xy": [[0.2, 0.2], [0.2, 0.8], [1.8, 0.2], [1.0, 0.9]]},
    ]
    batch = NpvBatch(gm, DT, NT)
    values = batch(params)
    ref = np.array([oracle_npv(om, DT, NT, np.zeros(n * n), **p)[0] for p in params])
    assert ref[5] == 0 and ref[6] == 0 and values[5] == 0 and values[6] == 0
    assert np.all(np.abs(ref[[0, 1, 2, 3, 4, 7]]) > 1)
    np.testing.assert_allclose(values, ref, rtol=1e-7, atol=1e-7)
    # the invalid members were flagged by the device as well (no flow -> no CFL step)
    assert batch.last["status"][5] != 0 and batch.last["status"][6] != 0
    # the device plan is reused by the next call of the same batch shape (an EnOpt iteration): steady rates only -> one
    # source-field column per member this time
    again = batch([params[k] for k in (7, 3, 2, 1, 0, 5, 6, 0)])
    np.testing.assert_allclose(again, ref[[7, 3, 2, 1, 0, 5, 6, 0]], rtol=1e-7, atol=1e-7)


def test_npv_batch_members_with_different_numbers_of_wells():
    """Optimise.py:736-767 varies the wells per member; one batch may hold members with two, three or four producers and one or two
    injectors (the device sees a source field per member): values equal the per-member oracle, and the values each member gets in a
    batch of its own kind."""
    from historymatching_amd.opt import NpvBatch
    from oracle.opt import npv as oracle_npv

    n = 20
    om, gm = _models(n)
    params = [
        {},
        {"prd_xy": [[0.2, 0.2], [1.8, 0.8], [1.0, 0.9]], "prd_rates": 1.5 * np.array([[0.5], [0.3], [0.2]])},
        {"prd_xy": [[0.2, 0.8], [1.8, 0.2]], "prd_rates": 1.5 * np.array([[0.5], [0.5]])},
        {"inj_xy": [[0.6, 0.5], [1.4, 0.5]], "inj_rates": 1.5 * np.array([[0.7], [0.3]])},
        {"inj_xy": [[0.6, 0.5], [1.4, 0.5]], "inj_rates": 1.5 * np.array([[0.5], [0.5]]),
         "prd_xy": [[0.1, 0.1], [1.9, 0.9], [1.0, 0.1]], "prd_rates": 1.5 * np.ones((3, 1)) / 3},
    ]
    batch = NpvBatch(gm, DT, NT)
    values = batch(params)
    ref = np.array([oracle_npv(om, DT, NT, np.zeros(n * n), **p)[0] for p in params])
    assert np.all(np.abs(ref) > 1) and len(set(np.round(ref, 6))) == len(ref)
    np.testing.assert_allclose(values, ref, rtol=1e-7, atol=1e-7)
    for k, p in enumerate(params):
        np.testing.assert_allclose(NpvBatch(gm, DT, NT)([p])[0], values[k], rtol=1e-12, atol=1e-12)


def test_npv_batch_one_permeability_per_member():
    """Robust objective: the same controls over an ensemble of permeability fields (Optimise.py:1006)."""
    from historymatching_amd.opt import NpvBatch
    from oracle.opt import npv as oracle_npv

    n, N = 20, 5
    om, gm = _models(n)
    Ks = 0.1 + np.exp(5 * perms(n, n, N, seed=5))
    ctrl = {"inj_xy": [[0.6, 0.4]]}
    values = NpvBatch(gm, DT, NT)([ctrl] * N, perms=Ks)
    ref = []
    for k in Ks:
        om.K = k
        ref.append(oracle_npv(om, DT, NT, np.zeros(n * n), **ctrl)[0])
    np.testing.assert_allclose(values, np.array(ref), rtol=1e-7)


@pytest.mark.parametrize("n,nT", [(128, 3), (256, 2)])
def test_npv_batch_large_grids_with_per_member_wells(n, nT):
    """At 128 x 128 the per-member source fields go through the matrix-core pressure solver (it only reads q) and the tiled
    saturation sweep (the register-resident kernels keep one shared well list and step aside); at 256 x 256 through the
    two-level CG pressure solver."""
    from historymatching_amd.opt import NpvBatch
    from oracle.opt import npv as oracle_npv

    om, gm = _models(n)
    params = [{}, {"inj_xy": [[0.31, 0.77]]}, {"inj_xy": [[-0.1, 0.5]]}, {"prd_rates": 1.5 * np.array([[0.4], [0.3], [0.2], [0.1]])}]
    values = NpvBatch(gm, DT, nT)(params)
    ref = np.array([oracle_npv(om, DT, nT, np.zeros(n * n), **p)[0] for p in params])
    assert ref[2] == 0 and values[2] == 0
    np.testing.assert_allclose(values, ref, rtol=1e-7, atol=1e-7)
