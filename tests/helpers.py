"""Shared builders for the parity tests (oracle side and GPU side use identical inputs)."""
import numpy as np

from historymatching_amd.geostat import gaussian_fields_kron


def wells_4corners(model):
    """Well layout of the reference (HistoryMatch.py:177-190)."""
    near01 = np.array([0.12, 0.87])
    model.prd_xy = [[x, y] for y in model.Ly * near01 for x in model.Lx * near01]
    model.inj_xy = [[model.Lx / 2, model.Ly / 2]]
    model.inj_rates = [[1]]
    model.prd_rates = np.ones((4, 1)) / 4
    return model


def make_models(Nx, Ny, dtype=64):
    """(oracle model, GPU-backed model) with identical grid and wells."""
    from historymatching_amd.ressim import ResSim as GpuResSim
    from oracle.ressim import ResSim as OracleResSim

    return wells_4corners(OracleResSim(Nx, Ny, 2, 1)), wells_4corners(GpuResSim(Nx, Ny, 2, 1, dtype=dtype))


def perms(Nx, Ny, N, seed=1, scale=1.0):
    return scale * gaussian_fields_kron(Nx, Ny, 2, 1, N, r=0.8, seed=seed)


def oracle_sim_and_noise(om, x, dt, nTime, wsat0=None):
    """Oracle simulation of one member plus the oracle's OWN numerical noise: the same restatement with
    SuperLU's column ordering switched from COLAMD to NATURAL.  The TPFA system is ill-conditioned
    (transmissibilities up to ~1e6: T*eps*|p| ~ 1e-9 flux noise for any fp64 solver), and the saturation
    front amplifies it, so parity of a different direct solver can only be asked to within this spread."""
    import oracle.ressim as orc
    from scipy.sparse.linalg import spsolve

    orc.set_perm(om, x)
    w0 = np.zeros(om.Nxy) if wsat0 is None else wsat0
    ref = om.sim(dt, nTime, w0)
    orig = orc.spsolve
    orc.spsolve = lambda A, b: spsolve(A.tocsc(), b, permc_spec="NATURAL")
    try:
        ref2 = om.sim(dt, nTime, w0)
    finally:
        orc.spsolve = orig
    return ref, float(np.abs(ref2 - ref).max())
