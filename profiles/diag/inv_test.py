import sys, ctypes as C, numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..'))
from historymatching_amd import _lib
ctx = _lib.Context.get(0); lib = _lib.load()
rng = np.random.RandomState(0)
for n in (16, 32, 48, 96, 160, 256):
    A = rng.randn(40, n); G = A.T @ A
    W = np.empty((n, n))
    rc = lib.hm_debug_spd_inverse(ctx.handle, n, G.ctypes.data_as(C.POINTER(C.c_double)), 39.0, W.ctypes.data_as(C.POINTER(C.c_double)))
    ref = np.linalg.inv(G + 39.0 * np.eye(n))
    err = np.abs(W - ref)
    i, j = np.unravel_index(err.argmax(), err.shape)
    print(n, rc, "max err", err.max(), "at tile", i // 16, j // 16, "sym err", np.abs(W - W.T).max(), "resid", np.abs(W @ (G + 39 * np.eye(n)) - np.eye(n)).max())
    if n == 160:
        te = np.array([[err[16*r:16*r+16, 16*c:16*c+16].max() for c in range(n//16)] for r in range(n//16)])
        np.set_printoptions(linewidth=200, precision=1)
        print(te)
# the reference's own case (tests/golden): C = S^T S + (N-1) I with S = centred obs * decorr
from pathlib import Path
g = Path('/root/repo/tests/golden')
f1, f2, f3 = (np.load(g / k) for k in ("f1_rng_replay.npz", "f2_obs_error.npz", "f3_ens_update0.npz"))
Y = f3["obs_ens"] - f3["obs_ens"].mean(0); S = Y @ f2["decorr"]; G = S.T @ S; n = G.shape[0]; N = Y.shape[0]
W = np.empty((n, n))
rc = lib.hm_debug_spd_inverse(ctx.handle, n, np.ascontiguousarray(G).ctypes.data_as(C.POINTER(C.c_double)), float(N - 1), W.ctypes.data_as(C.POINTER(C.c_double)))
Cm = G + (N - 1) * np.eye(n)
ref = np.linalg.inv(Cm)
err = np.abs(W - ref)
print("fixture: n", n, "N", N, "rc", rc, "cond", np.linalg.cond(Cm), "max |ref|", np.abs(ref).max(), "max err", err.max(), "resid", np.abs(W @ Cm - np.eye(n)).max(), "ref resid", np.abs(ref @ Cm - np.eye(n)).max())
te = np.array([[err[16*r:16*r+16, 16*c:16*c+16].max() for c in range(n//16)] for r in range(n//16)])
print(te)
print("diag range", np.diag(Cm).min(), np.diag(Cm).max())
