import sys, ctypes as C, numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..'))
from historymatching_amd import _lib
ctx = _lib.Context.get(0); lib = _lib.load()
rng = np.random.RandomState(0)
n, N = 160, 1000
A = rng.randn(400, n); G = np.ascontiguousarray(A.T @ A); X = rng.randn(N, n)
A_T = np.empty((n, N), dtype=np.float32)
dp = C.POINTER(C.c_double)
rc = lib.hm_debug_ldl_gain(ctx.handle, n, N, G.ctypes.data_as(dp), 39.0, X.ctypes.data_as(dp), A_T.ctypes.data_as(C.POINTER(C.c_float)))
ref = X @ np.linalg.inv(G + 39.0 * np.eye(n))
print("rc", rc, "max rel err", np.abs(A_T.T - ref).max() / np.abs(ref).max())
