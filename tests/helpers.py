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
