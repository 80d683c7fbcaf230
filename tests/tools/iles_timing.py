"""Time of the device half of one ILES iterate (HistoryMatch.py:1007-1064 at the batched form of :802-804): `hm_iles_step` (one
Gauss-Newton step of every local domain's N x N weight matrix: LU solve with W, n_loc x n_loc Cholesky, three N x N x n_loc
products) and `hm_iles_compose` (x0 + W_b X0 for every domain), synthetic observations (no forward model in the timed region),
against the host twin's per-element pseudo-inverse + SVD on a sample of domains.

    python tests/tools/iles_timing.py [N=100] [grid=128] [domain=8] [reps=5]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from historymatching_amd.localization import rectangular_partitioning  # noqa: E402
from historymatching_amd.update import IlesPlan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dom = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
nTime, nPrd = 40, 4
n_obs = nTime * nPrd
M = n * n
rng = np.random.RandomState(1)
batches = rectangular_partitioning((n, n), (dom, dom))
B = len(batches)
# a taper that keeps roughly a quarter of the observations per domain (the four-producer layout: the nearest producer's series)
taper_b = np.zeros((B, n_obs))
for b in range(B):
    near = rng.randint(nPrd)
    taper_b[b, near::nPrd] = 0.3 + 0.7 * rng.rand(nTime)
    taper_b[b, (near + 1) % nPrd::nPrd] = 0.05 * rng.rand(nTime)
prior = rng.randn(N, M)
plan = IlesPlan(prior, batches, taper_b, cutoff=1e-2)
S = rng.randn(N, n_obs)
S -= S.mean(0)
D = rng.randn(N, n_obs)
plan.step(S, D, 0.5)
plan.compose()
ts, tc = [], []
for _ in range(reps):
    t0 = time.perf_counter()
    plan.step(S, D, 0.5)
    t1 = time.perf_counter()
    plan.compose()
    t2 = time.perf_counter()
    ts.append(t1 - t0)
    tc.append(t2 - t1)
n_loc = int((np.sqrt(taper_b) > 1e-2).sum(1).mean())
flops = B * (2.0 * N * N * n_loc * 3 + 2.0 / 3 * N ** 3 + 2.0 * N * N * n_loc)  # three products, the LU of W, elimination + back substitution of the n_loc right-hand sides
print(f"ILES device step: N = {N}, state {n} x {n}, {B} domains of {dom} x {dom} cells, {n_obs} observations ({n_loc} in range per domain on average)")
print(f"   hm_iles_step    {1e3 * min(ts):8.2f} ms  (host call incl. the copy of S, D in; ~{flops / min(ts) / 1e9:.0f} GFLOP/s fp64 of the per-domain algebra)")
print(f"   hm_iles_compose {1e3 * min(tc):8.2f} ms  (incl. the copy of the {N} x {M} fp64 ensemble out: {N * M * 8 / 1e6:.0f} MB)")
plan.close()

# the reference's evaluation order on the host for a sample of domains: pinv(W) and svd(Y0) per domain
import scipy.linalg as sla  # noqa: E402

sample = min(B, 16)
W = np.eye(N) + 0.01 * rng.randn(N, N)
t0 = time.perf_counter()
for b in range(sample):
    c = np.sqrt(taper_b[b])
    jj = c > 1e-2
    Winv = sla.pinv(W)
    Y0 = (Winv - Winv.mean(0)) @ (S[:, jj] * c[jj])
    sla.svd(Y0, full_matrices=False)
host = (time.perf_counter() - t0) / sample * B
print(f"   host twin (pinv + svd per domain, {sample} domains timed, scaled to {B}): {1e3 * host:8.1f} ms on this box's cores")
